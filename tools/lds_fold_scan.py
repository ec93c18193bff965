"""ISA scan for the reduction shape behind round 3's reproducibility flake: every thread of a workgroup
writes partials to LDS, ONE workgroup barrier, then a FEW threads (a divergent region: exec narrowed by a compare) read many
of those words back and fold them.  Round 3's fused GroupNorm forward had exactly this shape and returned sums with partials
missing when the GPU was oversubscribed by 7 - 9 processes; the mechanism below the ISA was never isolated (the code is
race-free as written), so the defence is not to have the shape -- or to know where it is and stress it.

For every kernel of a gfx950 assembly file (hipcc -S --cuda-device-only) the scanner reports each
    ds_write ... s_barrier ... <exec narrowed> ... N >= MIN_READS ds_read ... <exec restored>
occurrence: (kernel, line of the barrier, LDS reads inside the divergent region; -1 = the reads sit in a loop, i.e. a serial
fold of unknown length: round 3's GroupNorm).  Uniform folds (all lanes read), butterflies
(ds_swizzle / DPP / ds_bpermute, no barrier) and wave-level folds do not match.
usage: python tools/lds_fold_scan.py file.s [...]      (exit code 0; one line per hit)"""
import re
import sys

MIN_READS = 4
KERNEL = re.compile(r"^(_Z\w+):\s*(;.*)?$")
NARROW = re.compile(r"^\s*(s_and_saveexec_b64|v_cmpx_\w+|s_and_b64\s+exec)")
RESTORE = re.compile(r"^\s*(s_or_b64\s+exec|s_mov_b64\s+exec)")


def scan(path):
    hits = []
    kernel, wrote, barrier_line = None, False, None
    lines = open(path).read().splitlines()
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = KERNEL.match(ln)
        if m:
            kernel, wrote, barrier_line = m.group(1), False, None
        elif kernel:
            s = ln.strip()
            if s.startswith("ds_write") or s.startswith("ds_store"):
                wrote = True
            elif s.startswith("s_barrier") and wrote:
                barrier_line = i + 1
            elif s.startswith("s_endpgm"):
                wrote, barrier_line = False, None
            elif barrier_line is not None and NARROW.match(ln):
                # divergent region: count LDS reads until exec is restored (or 400 lines / the next barrier)
                reads, j, labels, read_at, looped = 0, i + 1, {}, [], False
                sv = re.match(r"^\s*s_and_saveexec_b64\s+(s\[\d+:\d+\])", ln)
                saved = sv.group(1) if sv else None      # the region ends where THIS saved mask is restored
                while j < len(lines) and j < i + 400:
                    t = lines[j].strip()
                    if t.startswith("s_barrier") or t.startswith("s_endpgm"):
                        break
                    if RESTORE.match(lines[j]) and (saved is None or t.replace(" ", "").endswith("," + saved)):
                        break
                    lab = re.match(r"^(\.LBB\w+):", t)
                    if lab:
                        labels[lab.group(1)] = j
                    if t.startswith("ds_read") or t.startswith("ds_load"):
                        # ds_read2 / b64 / b128 read several words
                        reads += 2 if "read2" in t else 1
                        read_at.append(j)
                    br = re.match(r"^s_cbranch_\w+\s+(\.LBB\w+)", t)
                    if br and br.group(1) in labels and any(labels[br.group(1)] < r < j for r in read_at):
                        looped = True   # a backward branch over LDS reads: a serial fold loop (round 3's GroupNorm: 128 - 256 trips)
                        break
                    j += 1
                if looped or reads >= MIN_READS:
                    hits.append((kernel, barrier_line, -1 if looped else reads))
                    barrier_line = None   # one report per barrier
        i += 1
    return hits


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        for k, ln, n in scan(p):
            total += 1
            print(f"{p}: {k}: barrier at line {ln}, divergent fold " + ("in a LOOP of LDS reads" if n < 0 else f"with {n} LDS reads"))
    print(f"total {total}")

"""Generate the entry-point table of INTEGRATION.md's appendix from include/indm_hip.h: every exported function, the
reference lines its header comment cites, and the source file that defines it.
usage: python tools/abi_table.py            (prints markdown)"""
import glob, os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = open(os.path.join(ROOT, "include", "indm_hip.h")).read()
src = {os.path.basename(f): open(f).read() for f in glob.glob(os.path.join(ROOT, "indm_amd", "csrc", "*.hip"))}
# split the header into (comment, declarations that follow it up to the next comment)
parts = re.split(r"(/\*.*?\*/)", h, flags=re.S)
rows, comment = [], ""
REF = re.compile(r"[A-Za-z0-9_/\.]+\.(?:py|cpp|cu):\d+(?:-\d+)?(?:,\s?:?\d+(?:-\d+)?)*")
for part in parts:
    if part.startswith("/*"):
        comment = " ".join(l.strip(" */") for l in part.splitlines())
        continue
    for name in re.findall(r"\b(indm_[a-z0-9_]+)\s*\(", part):
        refs = REF.findall(comment)
        where = [f for f, t in src.items() if re.search(r"INDM_API\s+[\w\s\*]+?\b" + name + r"\s*\(", t)]
        rows.append((name, "; ".join(dict.fromkeys(refs)) or "— (companion of the entries around it: see the header comment)", ", ".join(sorted(where)) or "-"))
seen = set()
print("| entry point | reference lines its header comment names | defined in |")
print("|---|---|---|")
for name, refs, where in rows:
    if name in seen:
        continue
    seen.add(name)
    print(f"| `{name}` | {refs} | `{where}` |")

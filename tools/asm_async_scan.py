"""ISA check for the inline-asm asynchronous loads (global_load_dword* issued from an asm statement and awaited by a
LATER asm `s_waitcnt vmcnt`): the compiler believes the destination registers hold their value as soon as the asm
statement has issued, so it may copy, spill or read them before the wait -- a read of a register whose load is still in
flight returns stale data whenever memory is slower than the code in between (e.g. under co-tenancy).
Scans `hipcc -S` output: for every kernel, tracks the destination VGPRs of asm-issued global loads and reports any
instruction outside an asm block that READS one of them before the next asm `s_waitcnt vmcnt(...)`.
usage: asm_async_scan.py file.s [...]"""
import re
import sys


def regs(tok):
    """'v12' -> {12}; 'v[4:7]' -> {4,5,6,7}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def scan(path):
    fn, inasm, pending, hits = None, False, {}, []
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip() if not line.strip().startswith(";;#") else line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            fn, pending = m.group(1), {}
            continue
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ops = re.split(r"[\s,]+", t)
        mn, args = ops[0], ops[1:]
        if inasm:
            if mn.startswith("global_load_dword") and "lds" not in mn and args:
                for r in regs(args[0]):
                    pending[r] = ln
            elif mn == "s_waitcnt" and "vmcnt" in t:
                pending = {}
            continue
        if mn == "s_waitcnt" and re.search(r"vmcnt\(0\)", t):
            pending = {}   # a compiler-inserted full wait also retires them
            continue
        if not pending:
            continue
        # operand 0 is the destination for VALU / loads; stores and ds_write read every operand
        reads = args if (mn.startswith(("global_store", "buffer_store", "ds_write", "scratch_store", "flat_store"))) else args[1:]
        used = set()
        for a in reads:
            used |= regs(a)
        bad = used & set(pending)
        if bad:
            hits.append((fn, ln, t, sorted(bad)[:4], pending[sorted(bad)[0]]))
    return hits


if __name__ == "__main__":
    total = 0
    for f in sys.argv[1:]:
        hs = scan(f)
        total += len(hs)
        seen = {}
        for fn, ln, t, bad, src in hs:
            seen.setdefault(fn, []).append((ln, t, bad, src))
        for fn, v in seen.items():
            print(f"{f}: {fn[:100]}: {len(v)} reads of in-flight registers; first: {v[:3]}")
    print("total", total)
    sys.exit(1 if total else 0)

"""How deep are a kernel's loads pipelined in the ISA hipcc produced?  For every kernel of a `hipcc -S` file: the sequence of
global loads, `s_waitcnt vmcnt(N)` and MFMAs, reduced to
  * serial_loads : loads that are followed by `s_waitcnt vmcnt(0)` before any other load is issued (one memory latency each),
  * max_burst    : the longest run of loads with no vmcnt wait in between,
  * mfma_waits   : the vmcnt values of the waits that directly precede an MFMA group,
  * ring_moves   : v_mov_b64 / v_mov_b32 inside the region between the first and the last MFMA (register rotation).
Round 4 found two F16 kernels whose source pipelined their loads and whose ISA did not; tests/test_isa_cpu.py
keeps them pipelined.  usage: load_pipeline_scan.py file.s [kernel substring]"""
import re
import sys


def kernels(path):
    text = open(path).read()
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n", text):
        end = text.find(".Lfunc_end", m.end())
        yield m.group(1), text[m.end():end].splitlines()


def scan_body(lines):
    seq = []
    for ln in lines:
        t = ln.strip()
        if re.match(r"(global|buffer)_load_(dword|short|ubyte)", t) and " lds" not in t and "_lds_" not in t:
            seq.append(("L", None))
        elif t.startswith("s_waitcnt") and "vmcnt(" in t:
            seq.append(("W", int(re.search(r"vmcnt\((\d+)\)", t).group(1))))
        elif t.startswith("v_mfma"):
            seq.append(("M", None))
        elif t.startswith(("v_mov_b64", "v_mov_b32")):
            seq.append(("V", None))
    serial = burst = max_burst = 0
    mfma_waits, last_wait = [], None
    first_m = next((i for i, (k, _) in enumerate(seq) if k == "M"), None)
    last_m = max((i for i, (k, _) in enumerate(seq) if k == "M"), default=None)
    moves = sum(1 for i, (k, _) in enumerate(seq) if k == "V" and first_m is not None and first_m < i < last_m)
    prev = None
    for k, v in seq:
        if k == "L":
            burst += 1
            max_burst = max(max_burst, burst)
        elif k == "W":
            if v == 0 and prev == "L" and burst == 1:
                serial += 1
            burst = 0
            last_wait = v
        elif k == "M":
            if last_wait is not None:
                mfma_waits.append(last_wait)
                last_wait = None
        if k != "V":
            prev = k
    return {"serial_loads": serial, "max_burst": max_burst, "mfma_waits": mfma_waits, "ring_moves": moves}


def scan(path, sub=""):
    return {name: scan_body(body) for name, body in kernels(path) if sub in name}


if __name__ == "__main__":
    for name, r in scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "").items():
        print(name[:90], r)

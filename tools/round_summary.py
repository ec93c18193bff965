"""Print the numbers of a measurement set (tools/measure_round.sh output under gpurun_out/) for DESIGN.md section 7.
usage: python tools/round_summary.py"""
import csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def stats(d):
    fs = sorted(glob.glob(os.path.join(G, d, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    return fs[-1] if fs else None


for tag, fn in (("fp32", "final_bench.json"), ("bf16", "final_bench_bf16.json")):
    p = os.path.join(G, fn)
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    r = d["roofline"]
    print(f"== {tag}: {d['value']:.1f} img/s, {d['ms_per_step']:.1f} ms/step (median {d['ms_per_step_median']:.1f}), n_bar {d['config'].get('n_bar')}")
    print(f"   dominant {r['kernel']}: {r['avg_launch_ms'] * 1e3:.1f} us x {r['launches']} launches, {r['achieved']:.1f} TF = {r['frac']:.3f}; "
          f"alg bytes {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB; traffic {r['traffic']}")
    print(f"   whole step: {r['step']}")
    for k in ("cpu_baseline", "cpu_baseline_8_threads"):
        if d.get(k):
            print(f"   {k}: {d[k]['value']} img/s on {d[k]['cores']} threads")
    if d.get("nelbo"):
        n = d["nelbo"]
        print(f"   nelbo: max|d| {n.get('max_abs_diff_bpd')} bpd, {n['images_per_s']:.1f} img/s, {n.get('gflop_per_image', 0):.0f} GFLOP/img, frac {n.get('frac', 0):.3f}")
    if d.get("sampling") and "hipgraph" in d["sampling"]:
        s = d["sampling"]
        print(f"   sampling: {s['hipgraph']['gpu_ms_per_step']:.1f} ms/step at batch 256 (frac {s.get('frac', 0):.3f}), shard 32: "
              f"{s['per_rank_shard_of_8']['hipgraph']['gpu_ms_per_step']:.1f} ms/step")
    if d.get("bf16"):
        b = d["bf16"]
        print(f"   bf16 sub-leg: {b['value']:.1f} img/s, {b['ms_per_step']:.1f} ms/step; dominant {b['roofline']['avg_launch_ms'] * 1e3:.1f} us; step {b['roofline']['step']}")
    if d.get("train_per_rank_shard_of_8"):
        for k, v in d["train_per_rank_shard_of_8"].items():
            if isinstance(v, dict):
                print(f"   shard {k}: {v['ms_per_step']:.1f} ms/step, {v['images_per_s']:.0f} img/s")
for tag, d, steps in (("fp32", "final_prof", 4), ("bf16", "final_prof_bf16", 4)):
    f = stats(d)
    if f:
        print(f"== kernel classes {tag} ({os.path.relpath(f, ROOT)})")
        print(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_classes.py"), f, str(steps)], capture_output=True, text=True).stdout)
p = os.path.join(G, "final_config_bench.txt")
if os.path.exists(p):
    print(open(p).read())

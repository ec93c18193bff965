"""Bin a rocprofv3 *kernel_stats.csv (of `bench.py --steps S`) into kernel classes: ms per step and share.
usage: python tools/kernel_classes.py <kernel_stats.csv> <steps-in-the-profile (timed + warm-up)>"""
import csv
import sys

path, steps = sys.argv[1], float(sys.argv[2])
CLASSES = [
    ("H16 score net: conv / dgrad / 1x1", lambda n: "h16_conv_kernel" in n),
    ("H16 score net: weight grads", lambda n: "h16_wgrad_kernel" in n),
    ("H16 score net: split-K reduce", lambda n: "h16_reduce_kernel" in n),
    ("H16 score net: GroupNorm", lambda n: "h16_gn_" in n),
    ("H16 score net: bias-grad sums", lambda n: "h16_col" in n),
    ("H16 score net: pack / convert / resample", lambda n: "h16_" in n),
    ("F16 flow: series iteration in one launch (conv + gate + chain)", lambda n: "f16_chain_conv_kernel" in n),
    ("F16 flow: 64-row GEMM / chain / thin-N", lambda n: "f16_rows_kernel" in n),
    ("F16 flow: thin conv", lambda n: "f16_thin_conv" in n),
    ("F16 flow: weight grads", lambda n: "f16_thin_wgrad" in n or "f16_wgrad" in n),
    ("F16 flow: Sin-level backward", lambda n: "f16_bwd_act" in n),
    ("F16 flow: converters", lambda n: "f16_pack" in n or "f16_unpack" in n),
    ("mfma: glds gated", lambda n: "igemm_glds_kernel" in n and ", true," in n),
    ("mfma: glds", lambda n: "igemm_glds_kernel" in n),
    ("mfma: big tile", lambda n: "igemm_big_kernel" in n),
    ("mfma: b2b / fused flow", lambda n: "b2b" in n or "ires_chain" in n),
    ("mfma: attention", lambda n: "attn" in n),
    ("mfma: generic tiles", lambda n: "igemm_kernel" in n),
    ("thin (HBM-bound contractions)", lambda n: "thin" in n),
    ("split-K reduce", lambda n: "splitk_reduce" in n),
    ("elementwise: flow (sincos/ires/mul)", lambda n: any(k in n for k in ("sincos", "ires_", "indm_mul", "tap_", "axpby", "scale_rows"))),
    ("norm / softmax", lambda n: any(k in n for k in ("gn_", "bn_", "softmax"))),
    ("reductions", lambda n: any(k in n for k in ("colreduce", "segreduce", "rowreduce", "sqnorm"))),
    ("optimizer", lambda n: "adamw" in n or "ema" in n or "multi_copy" in n or "lop_" in n),
    ("ATen / copies", lambda n: "at::native" in n or "rocclr" in n or "hipMemcpy" in n.lower()),
]
tot = {}
calls = {}
for r in csv.DictReader(open(path)):
    n = r["Name"]
    ns = float(r["TotalDurationNs"])
    for cname, pred in CLASSES:
        if pred(n):
            break
    else:
        cname = "other HIP"
    tot[cname] = tot.get(cname, 0.0) + ns
    calls[cname] = calls.get(cname, 0) + int(r["Calls"])
total = sum(tot.values())
print(f"kernel time {total / steps / 1e6:.1f} ms/step over {sum(calls.values()) / steps:.0f} launches/step")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {k:40s} {v / steps / 1e6:8.2f} ms/step  {100 * v / total:5.1f} %  {calls[k] / steps:7.0f} launches/step")

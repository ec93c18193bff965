"""Probe of the im2col thin convolution (3 -> 512 channels at 32 x 32, batch 128: the flow's `t3 = C3^T v` and conv1): sweeps the
experiment knobs of igemm_thin_kflat_kernel<true, *> (INDM_THIN_TUNE, read ONCE per process: one setting per run of this script) on the GPU box.
bits 0-7: start delay of every second workgroup (x 1024 cycles); bit 8: non-temporal stores; bit 9: uncounted waits; bits 13-14: 32-column blocks per MFMA group 2 / 1 / 4;
bits 10-11: workgroups per CU (0 = default); bit 12: the round-4 LDS-staged kernel instead of thin_conv3_rows_kernel (whose only
knob is bits 10-11)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from indm_amd import hip

dev = "cuda"
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
CIN, HW = (int(v) for v in os.environ.get("KFLAT_GEOM", "3,32").split(","))   # 3,32 (scale 0) | 12,16 (scale 1) | 12,32 (CelebA)
x3, w3 = torch.randn(B, HW, HW, CIN, device=dev), torch.randn(3, 3, CIN, 512, device=dev)
bias, cb = torch.randn(512, device=dev), torch.randn(B, 512, device=dev)

def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

if len(sys.argv) > 3 and sys.argv[2] == "--one":   # PMC mode: a few launches of one variant (tools/pmc_kflat.sh)
    for _ in range(6):   # (INDM_THIN_TUNE comes from the caller's environment)
        hip.conv2d(x3, w3)
    torch.cuda.synchronize()
    sys.exit(0)

if len(sys.argv) > 3 and sys.argv[2] == "--tune":   # one value per process: the library reads INDM_THIN_TUNE once
    tune = int(sys.argv[3])
    y = hip.conv2d(x3, w3)
    torch.save(y.cpu(), f"/tmp/kflat_probe_{CIN}_{HW}_{tune}.pt")
    t0 = timeit(lambda: hip.conv2d(x3, w3))
    t1 = timeit(lambda: hip.conv2d(x3, w3, bias=bias, bias2=cb))
    ref = f"/tmp/kflat_probe_{CIN}_{HW}_4096.pt"
    ok = torch.equal(y.cpu(), torch.load(ref)) if os.path.exists(ref) else None
    print(f"tune {tune:5d} (wg/cu {(tune >> 10) & 3 or 'default'}, blocks per group code {(tune >> 13) & 3}, staged kernel {(tune >> 12) & 1}): "
          f"plain {t0:7.1f} us  {4.0 * B * HW * HW * 512 / t0 / 1e6:5.2f} TB/s   bias+bias2 {t1:7.1f} us   equal to the staged kernel: {ok}", flush=True)
    sys.exit(0)

import subprocess
for geom in ["3,32", "12,16", "12,32"]:
    print("geometry (channels, height = width):", geom, flush=True)
    for tune in [4096, 0, 32768, 0, 32768]:
        subprocess.run([sys.executable, __file__, str(B), "--tune", str(tune)],
                       env=dict(os.environ, INDM_THIN_TUNE=str(tune), KFLAT_GEOM=geom), check=True)

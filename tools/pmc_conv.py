"""Runs the two hottest kernels of the training step alone (bench.hot_kernel_cases: WaveNet In_l k=5 conv + gate, and its data gradient),
plus two calibration kernels of known traffic (a 1 GiB fill; a 1 GiB elementwise multiply that reads 1 GiB and writes 1 GiB - larger than
the 256 MiB Infinity Cache, so the reads come from HBM), for the rocprofv3 --pmc passes of tools/pmc_conv.sh."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B, T = int(os.environ.get("PMC_B", "32")), int(os.environ.get("PMC_T", "400"))
iters = int(os.environ.get("ITERS", "6"))
cases = bench.hot_kernel_cases("bf16", B, T)
x = torch.empty(256 << 20, device="cuda")           # 1 GiB of fp32
y = torch.empty_like(x)
for _ in range(iters):
    for c in cases.values():
        c["run"]()
    x.fill_(1.0)                                     # writes 1 GiB
    torch.mul(x, 1.5, out=y)                         # reads 1 GiB, writes 1 GiB
torch.cuda.synchronize()
print("ran", list(cases), "x", iters)

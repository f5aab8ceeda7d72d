"""Runs the dominant kernel (WaveNet In_i k=5 conv, gate epilogue) alone, for rocprofv3 --pmc passes and tuning."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import ops
B, T, H, k = 32, 400, 192, 5
R = B * (T + 4)
prec = ops.F32 if "f32" in sys.argv else ops.BF16
bf = prec == ops.BF16 and "f32a" not in sys.argv            # bf16 mode stores the WaveNet state / gates as bf16
a = torch.randn(R, H, device="cuda")
if bf:
    a = a.to(torch.bfloat16)
w = torch.randn(2 * H, H, k, device="cuda") / (H * k) ** 0.5
pw = ops.pack_weight(w, perm=ops.PERM_PAIR, perm_h=H, precision=prec)
bias = torch.zeros(2 * H, device="cuda")
G = torch.empty(R, 2 * H, device="cuda", dtype=a.dtype)
IO = (ops.IO_A_BF16 | ops.IO_OUT0_BF16) if bf else 0
ABL = int(os.environ.get("ABL", "0")) << 16
run = lambda: ops.conv_cl(a, pw, H, R, pad=2, epi=ops.EPI_GATE, flags=ABL, h=H, n=2 * H, rows_per_utt=T + 4, bias=bias, out0=G, ld0=2 * H, io_flags=IO)
n = int(os.environ.get("ITERS", "20"))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
print(f"ABL={ABL >> 16} conv In k=5 {'f32' if prec == ops.F32 else 'bf16'}: {us:.1f} us/launch, {2.0 * B * T * 2 * H * H * k / us / 1e6:.1f} TFLOP/s")

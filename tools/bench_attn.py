"""Times the relative-position attention core alone at the bench shape (B = 32 utterances x 2 heads, 120 tokens, D = 96), fp32 and bf16 mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd.conv_fn import RPRAttention
from glow_tts_amd import ops
B, T, H, D, win = int(os.environ.get("B", "32")), int(os.environ.get("T", "120")), 2, 96, 4
Tp = T + 4
g = torch.Generator().manual_seed(0)
rowmask = torch.zeros(B, Tp); rowmask[:, 2:T + 2] = 1
rowmask = rowmask.reshape(-1).cuda()
qkv = (torch.randn(B * Tp, 3 * H * D, generator=g) * 0.5).cuda().requires_grad_(True)
relk = (torch.randn(1, 2 * win + 1, D, generator=g) * 0.1).cuda().requires_grad_(True)
relv = (torch.randn(1, 2 * win + 1, D, generator=g) * 0.1).cuda().requires_grad_(True)
dout = torch.randn(B * Tp, H * D, generator=g).cuda()
for prec, name in ((ops.F32, "f32"), (ops.BF16, "bf16")):
    def fwd():
        return RPRAttention.apply(qkv, relk, relv, rowmask, B, Tp, H, win, 0.1, 7, None, prec)
    for _ in range(3):
        fwd().backward(dout)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 20
    tf = tb = 0.0
    for _ in range(n):
        e[0].record(); o = fwd(); e[1].record(); o.backward(dout); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"{name}: forward {tf / n * 1e3:.1f} us, backward {tb / n * 1e3:.1f} us (eager, includes the allocations and the colsum launch)")

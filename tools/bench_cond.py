"""GPU: the decoder's conditioning convs (decoder.CondLinear, csrc/cond_ops.hip) against torch's weight norm + einsum, forward and backward, us per call.
usage: python tools/bench_cond.py [B] [D]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glow_tts_amd.decoder import CondLinear, WeightNorm        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
v = (torch.randn(48, 384, D, 1, device="cuda") * 0.1).requires_grad_(True)
g = (torch.rand(48, 384, 1, 1, device="cuda") + 0.5).requires_grad_(True)
b = torch.randn(48, 384, device="cuda").requires_grad_(True)
vec = torch.randn(B, D, device="cuda").requires_grad_(True)
dout = torch.randn(B, 48 * 384, device="cuda")


def hip():
    return CondLinear.apply(g, v, b, vec)


def ref():
    w = WeightNorm.apply(g, v).squeeze(-1)
    return (torch.einsum("nod,bd->bno", w, vec) + b).view(B, -1)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, f in (("hip", hip), ("torch", ref)):
    with torch.no_grad():
        tf = timed(f)
    out = f()
    tb = timed(lambda: torch.autograd.grad(out, (g, v, b, vec), dout, retain_graph=True))
    print(f"B={B} D={D} {name:6s} forward {tf:7.1f} us   backward {tb:7.1f} us")

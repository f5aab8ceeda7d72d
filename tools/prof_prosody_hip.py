"""Kernel times of the prosody encoder's conv stack on the HIP path (B = 32, 80 x 800 mels): run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd.prosody import conv_stack_hip
prec = int(os.environ.get("PREC", "1"))
g = torch.Generator().manual_seed(0)
convs, cin = [], 1
for c in [32, 32, 64, 64, 128, 128]:
    conv = torch.nn.Conv2d(cin, c, 3, stride=2, padding=1, bias=False).cuda()
    convs.append(conv); cin = c
mels = torch.randn(32, 80, 800, generator=g).cuda()
for it in range(6):
    out = conv_stack_hip(convs, mels, prec)
    out.backward(torch.ones_like(out))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(5):
    out = conv_stack_hip(convs, mels, prec)
    out.backward(torch.ones_like(out))
e1.record(); torch.cuda.synchronize()
print("fwd+bwd per call (eager): %.1f us" % (e0.elapsed_time(e1) * 1e3 / 5))

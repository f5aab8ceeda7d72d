"""Where do NI = 1 and NI = 2 strips differ inside the model?  (tools build: GLOWTTS_LIB_PATH = tools/_build/libglowtts_hip_gemm.so)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import test_gpu_benchmarked_sizes as T
tl, ml = T._set_v(32, 5)
case = T.make_case("Vanilla", tl, ml, 2025)
from helpers import launch_counts, launch_reset
outs = {}
for v in ("1", "0", "1"):
    os.environ["GLOWTTS_DMA_NI1"] = v
    model = T._build("Vanilla", "bf16", case["sd"]).cuda().eval()
    c = lambda k: case[k].cuda()
    launch_reset()
    z, mm, ms, ld, dur, durt, attn, _ = model(c("tokens"), c("tl"), c("mels"), c("ml"), None, None, None)
    torch.cuda.synchronize()
    cur = dict(z=z.detach(), mm=mm.detach(), ms=ms.detach(), dur=dur.detach(), attn=attn.detach())
    if outs:
        for k in cur:
            d = (cur[k].float() - outs[k].float()).abs()
            print(f"NI1={v} vs first run: {k}: max diff {d.max().item():.3e}, elements differing {int((d > 0).sum())}")
    else:
        outs = cur
        print({k: n for k, n in launch_counts().items() if "conv" in k})

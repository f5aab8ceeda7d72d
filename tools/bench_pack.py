"""GPU: the text encoder's per-step weight packing (ops.PackSet.run) alone - the tile kernel over a device job table (glowtts_prep_launch_dev) against the
element-wise gather (glowtts_pack_weight_multi); us per launch, HIP events over a replayed graph of 20 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glow_tts_amd import ops          # noqa: E402
from glow_tts_amd import decoder      # noqa: E402,F401

torch.manual_seed(0)
ws = {}
for i in range(3):
    ws[f"pre{i}"] = torch.randn(192, 192, 5, device="cuda")
for i in range(6):
    ws[f"qkv{i}"] = torch.randn(576, 192, 1, device="cuda")
    ws[f"proj{i}"] = torch.randn(192, 192, 1, device="cuda")
    ws[f"c0_{i}"] = torch.randn(768, 192, 3, device="cuda")
    ws[f"c1_{i}"] = torch.randn(192, 768, 3, device="cuda")
ws["project"] = torch.randn(160, 192, 1, device="cuda")
ws["dp0"] = torch.randn(256, 192, 3, device="cuda")
ws["dp1"] = torch.randn(256, 256, 3, device="cuda")
items = [(k, w, tr) for k, w in ws.items() for tr in (False, True)]
for fast in (True, False):
    ops.FAST_PACK["on"] = fast
    ps = ops.PackSet(items, ops.BF16)
    for _ in range(3):
        ps.run()
    torch.cuda.synchronize()
    g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                ps.run()
        g.replay()
        e0.record(st)
        g.replay()
        e1.record(st)
    torch.cuda.synchronize()
    print(f"{'tile kernel (prep_dev)' if fast else 'element-wise gather  '}: {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us per launch, {len(items)} images, "
          f"{ps.data.numel() / 1e6:.1f} MB")

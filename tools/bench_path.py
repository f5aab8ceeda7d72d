"""GPU time of the dense-path kernel (glowtts_mas_path_from_idx) alone, replayed from a hipGraph (no host launch time in the figure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import monotonic_align as ma
for B in (32, 256):
    Tx, Ty = 120, 800
    idx = torch.randint(0, Tx, (B, Ty), device="cuda", dtype=torch.int32)
    for _ in range(3): ma.path_from_idx(idx, Tx)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): p = ma.path_from_idx(idx, Tx)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"B={B}: {us:.2f} us per launch, {4.0 * B * Tx * Ty / us / 1e6:.2f} TB/s")

"""Inference (GlowTTS.inference, Modules.py:128-204) latency / throughput at BASELINE-like sizes: eager launches and the two-graph
replay of glow_tts_amd.graph_infer.GraphedInference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
model, _, hp = bench.build_model("bf16", dev)
model.eval()
for B in (1, 8, 32):
    tokens, tl, mels, ml = bench.synthetic_batch(B, 120, 800, 80, 1, dev)
    model.train(); model(tokens, tl, mels, ml, None, None, None); model.eval()       # ActNorm init
    ls = torch.tensor([1.0], device=dev)
    with torch.no_grad():
        for _ in range(3):
            out, lengths, _ = model.inference(tokens, tl, None, None, None, None, None, None, noise_scale=0.667, length_scale=ls)
        torch.cuda.synchronize()
        t0 = time.time(); n = 10
        for _ in range(n):
            out, lengths, _ = model.inference(tokens, tl, None, None, None, None, None, None, noise_scale=0.667, length_scale=ls)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
    frames = int(lengths.sum())
    print(f"B={B}: {dt * 1e3:.2f} ms per batch, {frames} frames -> {frames / dt / 1e6:.3f} M frames/s, "
          f"RTF {dt / (frames * 256 / 22050):.5f} (256-hop, 22.05 kHz)")

from glow_tts_amd.graph_infer import GraphedInference
gi = GraphedInference(model)
for B in (1, 8, 32):
    tokens, tl, mels, ml = bench.synthetic_batch(B, 120, 800, 80, 1, dev)
    for _ in range(3):
        out, lengths, _ = gi(tokens, tl, noise_scale=0.667, length_scale=1.0)
    torch.cuda.synchronize()
    t0 = time.time(); n = 20
    for _ in range(n):
        out, lengths, _ = gi(tokens, tl, noise_scale=0.667, length_scale=1.0)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    frames = int(lengths.sum())
    print(f"graphed B={B}: {dt * 1e3:.2f} ms per batch, {frames} frames -> {frames / dt / 1e6:.3f} M frames/s, "
          f"RTF {dt / (frames * 256 / 22050):.5f}")

"""Training step at the reference's maximum data sizes (Train.Text_Length.Max = 200 tokens, Mel_Length.Max = 1000 frames), hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glow_tts_amd.graph_step import GraphedTrainStep
dev = torch.device("cuda:0")
model, mle, hp = bench.build_model("bf16", dev)
B, Tt, Tm = 32, 200, 1000
batch = bench.synthetic_batch(B, Tt, Tm, 80, 7, dev)
def loss_fn(m, tokens, tl, mels, ml):
    z, mm, ms, ld, dur, durt, _, _ = m(tokens, tl, mels, ml, None, None, None)
    return mle(z=z, mean=mm, std=ms, log_dets=ld, lengths=ml) + torch.nn.functional.mse_loss(dur, durt)
step = GraphedTrainStep(model, loss_fn)
for _ in range(3):
    loss = step(*batch)
torch.cuda.synchronize()
t0 = time.time(); n = 20
for _ in range(n):
    loss = step(*batch)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"B={B} tokens={Tt} frames={Tm}: {dt * 1e3:.2f} ms/step, {B * Tm / dt / 1e6:.2f} M mel-frames/s, loss {loss.item():.4f}")

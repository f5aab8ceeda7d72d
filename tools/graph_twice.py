"""Captures the forward+backward step of a bench config twice and times both graphs (is the first capture special?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glow_tts_amd import _lib
cfgid = int(os.environ.get("CFG", "5"))
cfg = bench.CONFIGS[cfgid]
dev = torch.device("cuda", 0)
model, mle_loss, hp = bench.build_model("bf16", dev, cfg["mode"], cfg["spk_type"])
B = cfg["batch"]
batch = bench.synthetic_batch(B, 120, 800, 80, 1234, dev)
cond = bench.conditioning_inputs(cfg, B, 1234, dev, hp)
side = torch.cuda.Stream()
def fb():
    mle, length = bench.forward_losses(model, mle_loss, batch, cond)
    model.zero_grad(set_to_none=True)
    (mle + length).backward()
    return (mle + length).detach()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(4):
        fb()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graphs, keep = [], []
for i in range(3):
    g = torch.cuda.CUDAGraph()
    with _lib.pinned_sink(keep):
        with torch.cuda.graph(g):
            out = fb()
    graphs.append(g); keep.append(out)
    for rep in range(2):
        g.replay(); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        print(f"graph {i} rep {rep}: {(time.time() - t0) / 20 * 1e3:.3f} ms/step")
for i, g in enumerate(graphs):
    g.replay(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"again graph {i}: {(time.time() - t0) / 20 * 1e3:.3f} ms/step")

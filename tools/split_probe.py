"""Experiment: does running two independent half-batch steps concurrently (two streams, two model replicas) beat one full-batch step?
Tests whether co-resident workgroups of independent kernels hide the per-kernel prologue / epilogue latency of the decoder chain."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 32
model, mle, hp = bench.build_model("bf16", dev)
models = [model] + [copy.deepcopy(model) for _ in range(NS - 1)]
for m in models[1:]:
    m._dec_stacks, m._enc_cache, m._enc_stream = None, {}, None
batches = [bench.synthetic_batch(B // NS, 120, 800, 80, 1234 + i, dev) for i in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
side = torch.cuda.Stream()
def step():
    cur = torch.cuda.current_stream()
    for s, m, b in zip(streams, models, batches):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            bench.train_step(m, mle, b)
    for s in streams:
        cur.wait_stream(s)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    step()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.time()
n = 20
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
print(f"splits={NS}: {(time.time() - t0) / n * 1e3:.3f} ms per {B}-utterance step")

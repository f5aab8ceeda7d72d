"""A/B: prosody encoder conv stack in NCHW vs channels_last (MIOpen inserts layout transposes around its NHWC kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
model, _, hp = bench.build_model("bf16", dev, "PE")
pe = model.layer_Dict["Prosody_Encoder"]
tokens, tl, mels, ml = bench.synthetic_batch(32, 120, 800, 80, 1, dev)

def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3

def stack(cl):
    def f():
        for p in pe.parameters(): p.grad = None
        x = mels.unsqueeze(1)
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        for i in range(pe.n_conv):
            x = pe.layer_Dict[f"Conv_{i}"](x)
        x.sum().backward()
    return f
print("conv stack fwd+bwd NCHW ms", t(stack(False)))
for i in range(pe.n_conv):
    c = pe.layer_Dict[f"Conv_{i}"].Conv
    c.weight.data = c.weight.data.contiguous(memory_format=torch.channels_last)
print("conv stack fwd+bwd channels_last ms", t(stack(True)))
g = torch.cuda.CUDAGraph()
f = stack(True)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    f(); f()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g): f()
print("channels_last graphed ms", t(g.replay))

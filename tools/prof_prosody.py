"""Where the PE-mode step spends its prosody-encoder time (plain torch modules): per-part eager timings at the bench shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
model, _, hp = bench.build_model("bf16", dev, "PE")
pe = model.layer_Dict["Prosody_Encoder"]
tokens, tl, mels, ml = bench.synthetic_batch(32, 120, 800, 80, 1, dev)

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3

def full():
    for p in pe.parameters(): p.grad = None
    pe(mels, ml).sum().backward()
print("prosody fwd+bwd eager ms", t(full))
with torch.no_grad():
    print("prosody fwd only ms", t(lambda: pe(mels, ml)))
    x = mels.unsqueeze(1)
    for i in range(pe.n_conv):
        blk = pe.layer_Dict[f"Conv_{i}"]
        print(f" conv {i} in {tuple(x.shape)} fwd ms", t(lambda: blk(x)))
        x = blk(x)
    xx = x.reshape(x.size(0), x.size(1) * x.size(2), x.size(3)).transpose(2, 1).contiguous()
    print(" gru in", tuple(xx.shape), "fwd ms", t(lambda: pe.layer_Dict["GRU"](xx)))
x = mels.unsqueeze(1)
for i in range(pe.n_conv):
    blk = pe.layer_Dict[f"Conv_{i}"]
    xi = x.detach().requires_grad_(i > 0)
    def fb():
        for p in blk.parameters(): p.grad = None
        blk(xi).sum().backward()
    print(f" conv {i} fwd+bwd ms", t(fb))
    x = blk(x).detach()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    full(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))

"""Which torch ops launch the elementwise / copy / fill glue kernels of one training step (eager mode, torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model, mle, hp = bench.build_model("bf16", dev)
batch = bench.synthetic_batch(32, 120, 800, 80, 1234, dev)
for _ in range(3):
    bench.train_step(model, mle, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    bench.train_step(model, mle, batch)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.device_time_total > 0 and e.key.startswith("aten::"):
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for t, c, k, s in rows[:45]:
    print(f"{t:9.1f} us {c:4d}  {k:28s} {s}")

"""Host-side cost of one eager training step (cProfile): where the Python / launch time goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
model, mle, hp = bench.build_model("bf16", dev)
batch = bench.synthetic_batch(32, 120, 800, 80, 1234, dev)
for _ in range(5):
    bench.train_step(model, mle, batch)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    bench.train_step(model, mle, batch)
t_host = (time.time() - t0) / 10
torch.cuda.synchronize()
print(f"host time per step (launch side only): {t_host * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    bench.train_step(model, mle, batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)

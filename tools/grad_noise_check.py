"""Is a 7e-3 (relative to the tensor's largest entry) deviation of an fp32 gradient from the fp32 oracle at B = 32 a defect or fp32 summation
noise?  Runs the oracle in float64 on the same case and prints, for the worst tensors, HIP-f32 vs oracle-f64 and oracle-f32 vs oracle-f64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import glowtts_ref as O
import test_gpu_benchmarked_sizes as T
tl, ml = T._set_v(32, 5)
case = T.make_case("Vanilla", tl, ml, 2025)
r = T.run_hip(case, "f32")
cfg = O.Cfg.from_yaml_dict(T._hp("Vanilla", "f32"))
sd64 = {k: (v.double() if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point()) for k, v in case["sd"].items()}
out = O.forward_train(sd64, cfg, case["tokens"], case["tl"], case["mels"].double(), case["ml"])
mle, length = O.train_losses(out, case["ml"], cfg)
(mle + length).backward()
rows = []
for k, want in case["grads"].items():
    w64 = sd64[k].grad
    s = w64.abs().max().item() + 1e-12
    rows.append(((r["grads"][k].double() - w64).abs().max().item() / s, (want.double() - w64).abs().max().item() / s, k))
rows.sort(reverse=True)
print("alignment equal to the f64 oracle:", torch.equal(out["attn"].float(), case["out"]["attn"]), "NLL f64", mle.item(), "f32 oracle", case["mle"], "HIP", r["mle"])
for a, b, k in rows[:8]:
    print(f"HIP-f32 vs oracle-f64 {a:.2e} | oracle-f32 vs oracle-f64 {b:.2e} | {k}")

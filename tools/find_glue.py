"""Which Python lines launch the PyTorch glue kernels of a training step (eager, torch.profiler with stacks): prints aten ops that launch a kernel, in order,
with the innermost frames of this package."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS[2] if hasattr(bench, "CONFIGS") else None
    model, mle_loss, hp = bench.build_model("bf16", dev, "Vanilla", None)
    batch = bench.synthetic_batch(32, 120, 800, 80, 1234, dev)
    from glow_tts_amd.optim import Modified_Noam_Scheduler, RAdam
    optimizer = RAdam(model.parameters(), lr=hp.Train.Learning_Rate.Initial, betas=(hp.Train.ADAM.Beta1, hp.Train.ADAM.Beta2), eps=hp.Train.ADAM.Epsilon)
    sched = Modified_Noam_Scheduler(optimizer, base=hp.Train.Learning_Rate.Base)
    opt = (optimizer, sched, hp.Train.Gradient_Norm)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            bench.train_step(model, mle_loss, batch, (None, None), None, 1, opt)
        torch.cuda.synchronize()
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            bench.train_step(model, mle_loss, batch, (None, None), None, 1, opt)
            torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.name.startswith("aten::") and e.device_type == torch.autograd.DeviceType.CPU and len(e.kernels) > 0]
    for e in evs:
        st = [s for s in (e.stack or []) if "glow_tts_amd" in s or "bench.py" in s][:3]
        print(e.name, [k.name[:60] for k in e.kernels][:2], "|", " <- ".join(s.split("/")[-1] for s in st))


main()

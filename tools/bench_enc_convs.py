"""The text encoder's conv shapes at the bench size (B = 32, 120 tokens -> 3 968 rows), LINEAR epilogue: fp32-stored A operand (register-staged
`conv_cl_kernel` / `conv_skinny_kernel`) against bf16-stored A (`conv_dma_kernel`).  Each shape is timed as a 50-launch hipGraph."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import launch_counts, launch_reset
from glow_tts_amd import ops

B, T = int(os.environ.get("ENC_B", "32")), int(os.environ.get("ENC_T", "120"))
R = B * (T + 4)
SHAPES = [("QKV 1x1", 192, 576, 1, 0), ("Projection 1x1", 192, 192, 1, ops.F_DROPOUT), ("Conv_0 k=3", 192, 768, 3, ops.F_RELU | ops.F_MASK | ops.F_DROPOUT),
          ("Conv_1 k=3", 768, 192, 3, ops.F_MASK | ops.F_DROPOUT), ("Conv_1^T k=3", 192, 768, 3, 0), ("Conv_0^T k=3", 768, 192, 3, 0),
          ("QKV^T 1x1", 576, 192, 1, 0), ("Prenet k=5", 192, 192, 5, 0), ("DP k=3", 192, 256, 3, ops.F_RELU | ops.F_MASK | ops.F_DROPOUT),
          ("DP^T k=3", 256, 192, 3, 0), ("Project 1x1", 192, 160, 1, ops.F_MASK), ("Project^T 1x1", 160, 192, 1, 0), ("DP0+spk k=3", 448, 256, 3, ops.F_RELU | ops.F_MASK | ops.F_DROPOUT)]
n = 50
st = torch.cuda.Stream()
rowmask = torch.ones(R, device="cuda")
seed_t = torch.tensor([5], device="cuda", dtype=torch.int32)
for name, ci, co, k, flags in SHAPES:
    w = torch.randn(co, ci, k, device="cuda") / (ci * k) ** 0.5
    pw = ops.pack_weight(w, precision=ops.BF16)
    bias = torch.zeros(co, device="cuda")
    a32 = torch.randn(R, ci, device="cuda")
    out = {}
    for label, a, io, odt in (("fp32 A", a32, 0, torch.float32), ("bf16 A", a32.to(torch.bfloat16), ops.IO_A_BF16, torch.float32),
                              ("bf16 A, bf16 out", a32.to(torch.bfloat16), ops.IO_A_BF16 | ops.IO_OUT0_BF16, torch.bfloat16)):
        y = torch.empty(R, co, device="cuda", dtype=odt)
        run = lambda: ops.conv_cl(a, pw, ci, R, lda=ci, pad=(k - 1) // 2, epi=ops.EPI_LINEAR, flags=flags | ops.F_BIAS, n=co, bias=bias, rowmask=rowmask,
                                  out0=y, ld0=co, drop_p=0.1 if flags & ops.F_DROPOUT else 0.0, seed=3, seed_t=seed_t, io_flags=io)
        launch_reset()
        run()
        which = [k_ for k_, v in launch_counts().items() if v][0]
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for _ in range(n):
                    run()
            g.replay()
            e0.record(st)
            g.replay()
            e1.record(st)
        torch.cuda.synchronize()
        out[label] = (e0.elapsed_time(e1) * 1e3 / n, which)
    fl = 2.0 * R * ci * co * k
    print(f"{name:16s} {ci:4d}->{co:4d}: " + " | ".join(f"{l} {us:6.1f} us ({fl / us / 1e6:5.0f} TF, {wh})" for l, (us, wh) in out.items()))

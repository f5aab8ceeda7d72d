"""conv_dma_kernel<LINEAR, 3> with 32- or 64-column strips (tools build: GLOWTTS_LIB_PATH = tools/_build/libglowtts_hip_gemm.so, GLOWTTS_DMA_NI1 = 1 / 0
- the switch is read once per process) against an fp64 reference on the bf16-rounded operands; prints a checksum to compare two runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import ops
torch.manual_seed(0)
B, T, k = 32, 120, 3
Tp = T + 4
R = B * Tp
for ci, co in [(768, 192), (192, 256), (256, 256), (256, 192), (192, 192)]:
    x = torch.randn(B, Tp, ci, device="cuda")
    x[:, :2] = 0; x[:, -2:] = 0
    a = x.reshape(R, ci).to(torch.bfloat16).contiguous()
    w = torch.randn(co, ci, k, device="cuda") / (ci * k) ** 0.5
    pw = ops.pack_weight(w, precision=ops.BF16)
    bias = torch.randn(co, device="cuda")
    ref = torch.nn.functional.conv1d(a.double().view(B, Tp, ci).transpose(1, 2), w.to(torch.bfloat16).double(), bias.double(), padding=1).transpose(1, 2).reshape(R, co)
    y = torch.full((R, co), 7.0, device="cuda")
    ops.conv_cl(a, pw, ci, R, lda=ci, pad=1, epi=ops.EPI_LINEAR, flags=ops.F_BIAS, n=co, bias=bias, out0=y, ld0=co, io_flags=ops.IO_A_BF16)
    torch.cuda.synchronize()
    err = (y.double() - ref).abs()
    colerr = err.amax(0)
    print(f"NI1={os.environ.get('GLOWTTS_DMA_NI1')} {ci}->{co}: max |y - fp64| {err.max().item():.3e} mean {err.mean().item():.3e} checksum {y.double().sum().item():.10e}; "
          f"worst columns {colerr.topk(4).indices.tolist()} errors {[f'{v:.1e}' for v in colerr.topk(4).values.tolist()]}")

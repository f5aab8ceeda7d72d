"""Times the decoder's LDS-DMA conv kernels alone at the bench shape (B = 32, T = 400): In_l forward / data gradient (bench.hot_kernel_cases)
plus the 1x1 Res_Skip conv and the two-source gate-derivative conv.  With GLOWTTS_LIB_PATH pointing at the tools build
(`make -C glow_tts_amd/csrc tools`) the GLOWTTS_* switches of csrc/tunable.h are live: used by tools/ab_conv.sh for A/B runs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from glow_tts_amd import ops

B, T, H = int(os.environ.get("PMC_B", "32")), int(os.environ.get("PMC_T", "400")), 192
R = B * (T + 4)
cases = bench.hot_kernel_cases("bf16", B, T)
dev = "cuda"
bf = torch.bfloat16
rowmask = torch.ones(R, device=dev)
# Res_Skip_l (not last): acts [R, H] bf16 -> res (h + rs)*mask bf16 [R, H], skip fp32 [R, H] accumulated
w_rs = torch.randn(2 * H, H, 1, device=dev) / H ** 0.5
pw_rs = ops.pack_weight(w_rs, precision=ops.BF16)
acts = torch.randn(R, H, device=dev).to(bf)
hin = torch.randn(R, H, device=dev).to(bf)
hout = torch.empty(R, H, device=dev, dtype=bf)
skip = torch.zeros(R, H, device=dev)
b_rs = torch.zeros(2 * H, device=dev)
cases["rs_1x1"] = dict(run=lambda: ops.conv_cl(acts, pw_rs, H, R, epi=ops.EPI_RESSKIP, flags=0, n=2 * H, h=H, rows_per_utt=T + 4, bias=b_rs, rowmask=rowmask,
                                               in0=hin, ldi0=H, out0=hout, ld0=H, out1=skip, ld1=H,
                                               io_flags=ops.IO_A_BF16 | ops.IO_IN0_BF16 | ops.IO_OUT0_BF16),
                       flops=2.0 * B * T * 2 * H * H)
# gate derivative: [d h_next | d skip] (K = 2H, two bf16 sources) x Res_Skip^T -> d acts -> (da, ds) PAIR-packed bf16 [R, 2H]
pw_rst = ops.pack_weight(w_rs, transpose=True, precision=ops.BF16)
dnext = torch.randn(R, H, device=dev).to(bf)
dskip = torch.randn(R, H, device=dev).to(bf)
gates = torch.rand(R, 2 * H, device=dev).to(bf)
dins = torch.empty(R, 2 * H, device=dev, dtype=bf)
cases["dgate_1x1"] = dict(run=lambda: ops.conv_cl(dnext, pw_rst, 2 * H, R, a2=dskip, lda2=H, ca1=H, epi=ops.EPI_DGATE, n=H, rows_per_utt=T + 4, rowmask=rowmask,
                                                  in0=gates, ldi0=2 * H, out0=dins, ld0=2 * H,
                                                  io_flags=ops.IO_A_BF16 | ops.IO_IN0_BF16 | ops.IO_OUT0_BF16),
                          flops=2.0 * B * T * 2 * H * H)
# Start conv (Modules.py:791): x_a = first 80 of 160 fp32 channels -> h0 bf16 [R, H]; and its data gradient accumulated into d x_a
C = 160
w_st = torch.randn(H, C // 2, 1, device=dev) / (C // 2) ** 0.5
pw_st, pw_stt = ops.pack_weight(w_st, precision=ops.BF16), ops.pack_weight(w_st, transpose=True, precision=ops.BF16)
xmid = torch.randn(R, C, device=dev)
h0 = torch.empty(R, H, device=dev, dtype=bf)
b_st = torch.zeros(H, device=dev)
cases["start_fwd"] = dict(run=lambda: ops.conv_cl(xmid, pw_st, C // 2, R, lda=C, epi=ops.EPI_LINEAR, flags=ops.F_BIAS | ops.F_MASK, n=H, rows_per_utt=T + 4, bias=b_st,
                                                  rowmask=rowmask, out0=h0, ld0=H, io_flags=ops.IO_OUT0_BF16), flops=2.0 * B * T * H * C // 2)
dh0 = torch.randn(R, H, device=dev)
dx = torch.zeros(R, C, device=dev)
cases["start_dgrad"] = dict(run=lambda: ops.conv_cl(dh0, pw_stt, H, R, lda=H, epi=ops.EPI_LINEAR, flags=ops.F_ACCUM, n=C // 2, rows_per_utt=T + 4, out0=dx, ld0=C),
                            flops=2.0 * B * T * H * C // 2)
out = []
for name, c in cases.items():
    sec = bench.time_kernel(c["run"], iters=int(os.environ.get("ITERS", "40")))
    out.append(f"{name} {sec * 1e6:6.2f} us ({c['flops'] / sec / 1e12:5.0f} TF/s)")
print(" | ".join(out))

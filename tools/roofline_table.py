"""Per-kernel roofline table of the training step: joins the rocprofv3 per-step kernel summary written by
tools/profile_step.sh (gpurun_out/<tag>_kernel_stats.csv) with the ALGORITHMIC flops / bytes per launch of each kernel class at the bench shape
(config 2: B = 32, 400 squeezed frames -> 12 800 valid rows, 120 tokens -> 3 840 token rows) and prints the top kernels as a markdown table:
us per launch, launches and us per step, TFLOP/s and fraction of the 2.5 PFLOP/s dense-bf16 MFMA peak, GB/s and fraction of 8 TB/s.
    python tools/roofline_table.py gpurun_out/r03_kernel_stats.csv > profiles/r03_kernel_roofline.md"""
import csv
import sys

R, RT = 32 * 400, 32 * 120                      # valid decoder rows / token rows
H, C, L, K = 192, 160, 4, 5
MB = 1e6
# (substring of the kernel name, label, flops per launch, algorithmic bytes per launch)
WN_ROW = 2.0 * (80 * H + L * H * 2 * H * K + (L - 1) * H * 2 * H + H * H + H * C)
TABLE = [
    ("wn_fwd_kernel", "fused coupling network forward (1 flow)", R * WN_ROW, R * (640 + 320 + L * 1536 + 768 + 640) + 3.59 * MB),
    ("wn_bwd_kernel", "fused coupling network data gradients (1 flow)", R * WN_ROW, R * (384 + L * (768 + 768 + 384) + 384 + 768 + 640) + 3.59 * MB),
    ("conv_dma_kernel<0, 5", "In_l data gradient k=5 384->192", 2.0 * R * 384 * 192 * 5, R * (768 + 384 + 384) + 0.74 * MB),
    ("conv_dma_kernel<1, 5", "In_l forward k=5 + gate 192->384", 2.0 * R * 384 * 192 * 5, R * (384 + 768 + 384) + 0.74 * MB),
    ("conv_dma_kernel<4, 1", "gate derivative 1x1 (2 sources) 384->192", 2.0 * R * 384 * 192, R * (768 + 768 + 768) + 0.15 * MB),
    ("conv_dma_kernel<2, 1", "Res_Skip_l 1x1 192->384", 2.0 * R * 384 * 192, R * (384 + 384 + 384 + 1536) + 0.15 * MB),
    ("conv_chain_kernel<0, 4", "End^T -> last gate derivative (chained)", 2.0 * R * (192 * 192 * 2), R * (384 + 384 + 768 + 768)),
    ("conv_chain_kernel<2, 3", "last Res_Skip -> End + coupling (chained)", 2.0 * R * (192 * 192 + 192 * 160), R * (384 + 768 * 2 + 640 * 2)),
    ("wgrad_kernel<bool _Accum, int, ELi0E, true, true, true>", "weight gradients, k=5 group (48 problems)", 48 * 2.0 * R * 384 * 192 * 5, 48 * R * (768 + 384)),
    ("wgrad_kernel<bool _Accum, int, E, 0, true, true, true>", "weight gradients, Res_Skip 1x1 group", 12 * 7 * 2.0 * R * 192 * 192, 12 * 7 * R * 768),
    ("wgrad_kernel<bool _Accum, int, E, 0, false, false, true>", "weight gradients, Start / End group (fp32 operands)", 12 * 2.0 * R * (192 * 80 + 160 * 192), 12 * R * (768 + 320 + 768 + 768)),
    ("conv_cl_kernel<bool _Accum, int, E, 2, 4, 1, 0, 3", "encoder FFN conv k=3 (192<->768, fp32 rows)", 2.0 * RT * 768 * 192 * 3, RT * (768 + 3072) * 1.0),
    ("conv_dma_kernel<0, 3", "encoder k=3 convs on bf16 rows (FFN 192<->768 forward + data gradients, duration predictor; priced as FFN)", 2.0 * RT * 768 * 192 * 3, RT * (384 + 1536) * 1.0),
    ("conv_dma_kernel<0, 1", "encoder 1x1 convs on bf16 rows (QKV 192->576 and data gradients; priced as QKV)", 2.0 * RT * 576 * 192, RT * (384 + 2304) * 1.0),
    ("attn_bwd_mfma_kernel", "relative-position attention backward (1 layer)", 32 * 2 * 2.5 * 8.0 * 120 * 120 * 96, 32 * (120 * 576 * 4 * 2 + 2 * 120 * 120 * 4)),
    ("attn_fwd_mfma_kernel", "relative-position attention forward (1 layer)", 32 * 2 * 8.0 * 120 * 120 * 96, 32 * (120 * 576 * 4 + 120 * 192 * 4 + 2 * 120 * 120 * 4)),
    ("ln_bwd_kernel", "LayerNorm backward", 0.0, None),
    ("actnorm_inv_bwd4_kernel", "ActNorm + 1x1 backward (+ next coupling backward)", 0.0, 32 * 404 * (640 * 4 + 768 * 2)),
    ("pack_weight_kernel", "weight packing (fp32 -> bf16 tile order)", 0.0, None),
    ("radam_kernel", "RAdam update (28.6 M parameters)", 0.0, 28.6e6 * 4 * 7),
    ("multi_sqnorm_kernel", "gradient norm", 0.0, 28.6e6 * 4),
    ("mas_dp2_kernel", "MAS dynamic programme + backtrack (32 utterances, 120 x 800)", 0.0, 32 * 8 * 120 * 800),
]
PEAK_TF, PEAK_GB = 2500.0, 8000.0

rows = list(csv.DictReader(open(sys.argv[1])))
print(f"| kernel | launches / step | us / launch | us / step | TFLOP/s | frac MFMA peak | GB/s (algorithmic) | frac HBM |")
print("|---|---|---|---|---|---|---|---|")
tot = 0.0
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    name, calls, per_step, avg = r["Name"], float(r["CallsPerStep"]), float(r["TotalDurationNsPerStep"]) / 1e3, float(r["AverageNs"]) / 1e3
    tot += per_step
    hit = next((t for t in TABLE if t[0] in name), None)
    label = hit[1] if hit else name[:70]
    fl, by = (hit[2], hit[3]) if hit else (0.0, None)
    if hit and "wgrad_kernel" in hit[0] and calls > 1.5:        # the demangled name hides the tap count: the encoder's groups share the row
        label += " + the encoder's group of the same storage types (averaged: not priced)"
        fl, by = 0.0, None
    tf = fl / (avg * 1e-6) / 1e12 if fl else None
    gb = by / (avg * 1e-6) / 1e9 if by else None
    print(f"| {label} | {calls:.1f} | {avg:.1f} | {per_step:.0f} | " + (f"{tf:.0f} | {tf / PEAK_TF:.3f}" if tf else "- | -") + " | "
          + (f"{gb:.0f} | {gb / PEAK_GB:.3f}" if gb else "- | -") + " |")
print(f"\nserialised kernel time of the rows above: {tot / 1e3:.2f} ms per step; all kernels: {sum(float(r['TotalDurationNsPerStep']) for r in rows) / 1e6:.2f} ms per step "
      "(rocprofv3 runs a graph's kernels one at a time: the replayed step overlaps its two streams and is shorter).")

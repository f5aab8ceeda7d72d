"""In-kernel timeline of the dominant conv kernel (WaveNet In_i k=5, gate epilogue), built by tools/build_tl.sh.
Prints the kernel time and, per stamp, the median shader-clock offset from the workgroup's own start."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from glow_tts_amd import ops, _lib
B, T, H, k = 32, 400, 192, 5
R = B * (T + 4)
tl_lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libconv_tl.so"))
tl_lib.glowtts_conv_cl.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
bf = "f32a" not in sys.argv
a = torch.randn(R, H, device="cuda")
if bf:
    a = a.to(torch.bfloat16)
w = torch.randn(2 * H, H, k, device="cuda") / (H * k) ** 0.5
pw = ops.pack_weight(w, perm=ops.PERM_PAIR, perm_h=H, precision=ops.BF16)
bias = torch.zeros(2 * H, device="cuda")
G = torch.empty(R, 2 * H, device="cuda", dtype=torch.bfloat16 if bf else torch.float32)
NWG = 4096
TLS = int(os.environ.get("TL_STRIDE", "32"))
tl = torch.zeros(NWG, TLS, dtype=torch.int64, device="cuda")
args = ops.ConvArgs()
args.a, args.lda, args.ca, args.rows = a.data_ptr(), H, H, R
args.w, args.n, args.npad, args.kchunks, args.taps, args.pad, args.precision = pw.data.data_ptr(), 2 * H, pw.npad, pw.kchunks, 5, 2, ops.BF16
args.epi, args.flags, args.h, args.rows_per_utt = ops.EPI_GATE, int(os.environ.get("ABL", "0")) << 16, H, T + 4
args.bias, args.out0, args.ld0 = bias.data_ptr(), G.data_ptr(), 2 * H
args.ncols_valid = tl.data_ptr()
args.io_flags = (ops.IO_A_BF16 | ops.IO_OUT0_BF16) if bf else 0
run = lambda: _lib.check(tl_lib.glowtts_conv_cl(ctypes.byref(args), _lib.stream()), "conv")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    run()
e1.record(); torch.cuda.synchronize()
print(f"tile={os.environ.get('GLOWTTS_TILE', 'default')} dma={os.environ.get('GLOWTTS_DMA', '1')} waves={os.environ.get('GLOWTTS_DMA_WAVES', 'auto')} {e0.elapsed_time(e1) * 1e3 / n:.1f} us/launch")
if not int(os.environ.get("ABL", "0")):
    # correctness against a plain fp32 conv over the row axis (rows 2..R-3: the kernels differ in how they treat rows outside the tensor)
    import torch.nn.functional as F
    x = a.float().t().unsqueeze(0)                                    # [1, H, R]
    pre = F.conv1d(x, w.to(torch.bfloat16).float(), bias, padding=2)[0].t()           # [R, 2H]
    want = torch.stack([torch.tanh(pre[:, :H]), torch.sigmoid(pre[:, H:])], 2).reshape(R, 2 * H)
    err = (G.float() - want)[2:R - 2].abs().max().item()
    print(f"max |G - reference| = {err:.3e}")
    assert err < 2e-2
t = tl.cpu().numpy()
used = t[:, 0] != 0
t = t[used]
print("workgroups:", len(t))
t0 = t[:, 0].min()
rel = t[:, :30] - t[:, :1]
names = {0: "start", 1: "first loads issued", 2: "first tiles stored", 3: "sync"}
for ss in range(8):
    names[4 + 3 * ss] = f"ss{ss} compute done"; names[5 + 3 * ss] = f"ss{ss} sync"; names[6 + 3 * ss] = f"ss{ss} next tiles stored+sync"
if os.environ.get("GLOWTTS_DMA", "1") != "0" and bf:
    names = {0: "start", 1: "first chunk issued"}
    for ss in range(8):
        names[3 + 3 * ss] = f"ss{ss} landed + barrier"; names[4 + 3 * ss] = f"ss{ss} next chunk issued"; names[5 + 3 * ss] = f"ss{ss} compute done"
names[22] = "epilogue: biases loaded"; names[23] = "epilogue: 4 rows done"; names[24] = "epilogue: 8 rows done"; names[29] = "epilogue done"
prev = 0
for i in range(30):
    if t[0, i] == 0:
        continue
    med = np.median(rel[:, i])
    print(f"  [{i:2d}] {names.get(i, ''):32s} median +{med:8.0f} clk  (step {med - prev:7.0f})   p10 {np.percentile(rel[:, i], 10):7.0f}  p90 {np.percentile(rel[:, i], 90):7.0f}")
    prev = med
start = t[:, 0] - t0
end = t[:, 29] - t0
print(f"start offsets: median {np.median(start):.0f}  p90 {np.percentile(start, 90):.0f}  max {start.max():.0f};  end: median {np.median(end):.0f}  max {end.max():.0f} clk")
hw, xcc = t[:, 30], t[:, 31] & 0xF
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 16 + cu
u, c = np.unique(key, return_counts=True)
print(f"distinct (xcc,se,cu): {len(u)}; workgroups per CU: " + ", ".join(f"{v}x{(c == v).sum()}" for v in sorted(set(c))))

if TLS >= 128:
    print("per-wave stamps at chunk 3 (clk after the workgroup's start; SIMD id from HW_ID bits 4-5):")
    for w in range(16):
        col = 64 + 3 * w
        ok = t[:, col] != 0
        if not ok.any():
            continue
        a = np.median(t[ok, col] - t[ok, 0]); b = np.median(t[ok, col + 1] - t[ok, 0])
        simd = np.bincount(((t[ok, col + 2] >> 4) & 3).astype(int), minlength=4)
        print(f"  wave {w:2d}: barrier passed +{a:7.0f}  compute done +{b:7.0f}  (compute {b - a:6.0f})  simd histogram {simd.tolist()}")

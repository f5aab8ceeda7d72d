"""Does co-residency of two independent kernels hide the per-kernel latency?  The WaveNet In conv on the full batch on one stream
vs the two half batches on two streams, same total work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import ops
H, k, T = 192, 5, 400
def make(B):
    R = B * (T + 4)
    a = torch.randn(R, H, device="cuda").to(torch.bfloat16)
    w = torch.randn(2 * H, H, k, device="cuda") / (H * k) ** 0.5
    pw = ops.pack_weight(w, perm=ops.PERM_PAIR, perm_h=H, precision=ops.BF16)
    bias = torch.zeros(2 * H, device="cuda")
    G = torch.empty(R, 2 * H, device="cuda", dtype=torch.bfloat16)
    return lambda: ops.conv_cl(a, pw, H, R, pad=2, epi=ops.EPI_GATE, h=H, n=2 * H, rows_per_utt=T + 4, bias=bias, out0=G, ld0=2 * H,
                               io_flags=ops.IO_A_BF16 | ops.IO_OUT0_BF16)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
runs = [make(32 // NS) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
def burst(n):
    for s, r in zip(streams, runs):
        with torch.cuda.stream(s):
            for _ in range(n):
                r()
burst(5); torch.cuda.synchronize()
import time
t0 = time.time(); burst(200); torch.cuda.synchronize(); dt = time.time() - t0
print(f"streams={NS} stages={os.environ.get('GLOWTTS_DMA_STAGES', '3')}: {dt / 200 * 1e6:.1f} us per full-batch-equivalent launch")

"""Prices the fixed parts of the dominant conv kernel inside a hipGraph (no host launch cost): a chain of N launches of the In_l k=5 gate conv
that exit (256) at entry, (512) after the address set-up, (1024) after the first chunk has landed, or run completely (0).  Needs the
experiment build: hipcc ... -DGLOWTTS_TOOLS -DGLOWTTS_TOOLS_MIN -DGLOWTTS_ABL -> tools/_build/libconv_abl.so."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import ops, _lib
B, T, H, k = 32, 400, 192, 5
R = B * (T + 4)
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", os.environ.get("ABL_LIB", "libconv_abl.so")))
lib.glowtts_conv_cl.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
a = torch.randn(R, H, device="cuda").to(torch.bfloat16)
w = torch.randn(2 * H, H, k, device="cuda") / (H * k) ** 0.5
pw = ops.pack_weight(w, perm=ops.PERM_PAIR, perm_h=H, precision=ops.BF16)
bias = torch.zeros(2 * H, device="cuda")
G = torch.empty(R, 2 * H, device="cuda", dtype=torch.bfloat16)
def args_for(abl):
    args = ops.ConvArgs()
    args.a, args.lda, args.ca, args.rows = a.data_ptr(), H, H, R
    args.w, args.n, args.npad, args.kchunks, args.taps, args.pad, args.precision = pw.data.data_ptr(), 2 * H, pw.npad, pw.kchunks, 5, 2, ops.BF16
    args.epi, args.flags, args.h, args.rows_per_utt = ops.EPI_GATE, abl << 16, H, T + 4
    args.bias, args.out0, args.ld0 = bias.data_ptr(), G.data_ptr(), 2 * H
    args.io_flags = ops.IO_A_BF16 | ops.IO_OUT0_BF16
    return args
N = 200
for abl in (256, 512, 1024, 0):
    args = args_for(abl)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            assert lib.glowtts_conv_cl(ctypes.byref(args), ctypes.c_void_p(s.cuda_stream)) == 0
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(N):
            assert lib.glowtts_conv_cl(ctypes.byref(args), ctypes.c_void_p(st)) == 0
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"exit {abl:5d}: {e0.elapsed_time(e1) * 1e3 / (2 * N):6.2f} us per launch (graph replay, {N}-launch dependent chain)")

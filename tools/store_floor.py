"""What a dependent chain of STORE-only kernels costs per launch on the 8-XCD part: the same 12 MiB buffer rewritten vs 50 different buffers
(glowtts_fill_zero, hipGraph replay, no host time)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import _lib
L = _lib.lib()
L.glowtts_fill_zero.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
for mb in (1, 4, 12, 48, 192):
    n = mb * (1 << 20) // 4
    for nbuf in (1, 50):
        bufs = [torch.ones(n, device="cuda") for _ in range(nbuf)]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for b in bufs[:3]:
                L.glowtts_fill_zero(b.data_ptr(), n, _lib.stream())
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        N = 100
        with torch.cuda.graph(g):
            for i in range(N):
                L.glowtts_fill_zero(bufs[i % nbuf].data_ptr(), n, _lib.stream())
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (2 * N)
        print(f"{mb:4d} MiB per launch, {nbuf:2d} buffer(s): {us:6.2f} us per launch = {mb * 1.048576 / us:5.2f} TB/s")
        del bufs

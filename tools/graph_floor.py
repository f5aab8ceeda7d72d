"""Per-kernel floor of a dependent kernel chain replayed as a hipGraph (what a chain of N tiny launches costs on this platform)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import _lib
L = _lib.lib()
import ctypes
L.glowtts_fill_zero.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
x = torch.ones(1 << 20, device="cuda")
for n_elem in (64, 1 << 20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(10):
            L.glowtts_fill_zero(x.data_ptr(), n_elem, _lib.stream())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    N = 1000
    with torch.cuda.graph(g):
        for _ in range(N):
            L.glowtts_fill_zero(x.data_ptr(), n_elem, _lib.stream())
    g.replay(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print(f"{n_elem} floats per launch: {(time.time() - t0) / 5 / N * 1e6:.2f} us per dependent kernel in a graph")

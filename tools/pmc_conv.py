"""Runs the two hottest kernels of the training step alone (bench.hot_kernel_cases: WaveNet In_l k=5 conv + gate, and its data gradient),
plus two calibration kernels of known traffic (a 1 GiB fill; a 1 GiB elementwise multiply that reads 1 GiB and writes 1 GiB - larger than
the 256 MiB Infinity Cache, so the reads come from HBM), for the rocprofv3 --pmc passes of tools/pmc_conv.sh."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B, T = int(os.environ.get("PMC_B", "32")), int(os.environ.get("PMC_T", "400"))
iters = int(os.environ.get("ITERS", "6"))
cases = bench.hot_kernel_cases("bf16", B, T)
x = torch.empty(256 << 20, device="cuda")           # 1 GiB of fp32
y = torch.empty_like(x)
for _ in range(iters):
    for c in cases.values():
        c["run"]()
    x.fill_(1.0)                                     # writes 1 GiB
    torch.mul(x, 1.5, out=y)                         # reads 1 GiB, writes 1 GiB
if os.environ.get("PMC_PROBE"):                      # calibration of the MFMA-busy pass: a loop of back-to-back bf16 MFMAs, one wave per SIMD on every CU
    import ctypes
    from glow_tts_amd import _lib
    L = _lib.lib()
    L.glowtts_mfma_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    out = torch.zeros(ncu, 2, dtype=torch.int64, device="cuda")
    for _ in range(iters):
        _lib.check(L.glowtts_mfma_clock_probe(out.data_ptr(), ncu, 20000, None, _lib.stream()), "probe")
torch.cuda.synchronize()
print("ran", list(cases), "x", iters)
if os.environ.get("PMC_WN_BWD"):                    # + one flow's fused data-gradient launch (wn_bwd_kernel) at the same shape: tools/bench_wn.py's set-up
    import runpy
    os.environ.setdefault("ITERS", "6")
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_wn.py"), run_name="__main__")

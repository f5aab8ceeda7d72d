"""Settles the "sustained clock" question of VERDICT r2 (weak #8): does this MI355X hold 2.4 GHz under a dense bf16 MFMA load, or ~1.6-1.8?
Runs glowtts_mfma_clock_probe (one wave per SIMD on every CU, back-to-back v_mfma_f32_32x32x16_bf16 on random operands) for >= 100 ms per launch,
ten launches, while a thread samples `rocm-smi --showclocks --showpower`; prints the shader clock each launch measured (s_memtime cycles
against the constant-rate wall counter) next to the SMI samples.  Output is committed under profiles/.
    python tools/clock_log.py > gpurun_out/r03_clock_log.txt"""
import ctypes
import os
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import _lib

L = _lib.lib()
L.glowtts_mfma_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
ncu = torch.cuda.get_device_properties(0).multi_processor_count
out = torch.zeros(ncu, 2, dtype=torch.int64, device="cuda")
khz = ctypes.c_int(0)
ITERS = int(os.environ.get("ITERS", "1500000"))           # 4 MFMAs x 32 clocks per iteration: ~190 M clocks = ~100 ms at 1.8 GHz
samples, stop = [], False


def smi():
    while not stop:
        t = time.time()
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in r.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "fclk"))]
            samples.append((t, keep))
        except Exception as exc:                            # noqa: BLE001
            samples.append((t, [f"rocm-smi failed: {exc}"]))
        time.sleep(0.05)


th = threading.Thread(target=smi)
th.start()
time.sleep(0.5)
t_start = time.time()
rows = []
for i in range(10):
    t0 = time.time()
    _lib.check(L.glowtts_mfma_clock_probe(out.data_ptr(), ncu, ITERS, ctypes.byref(khz), _lib.stream()), "probe")
    torch.cuda.synchronize()
    t1 = time.time()
    o = out.double()
    ghz = (o[:, 0] / o[:, 1]) * khz.value * 1e-6
    rows.append((t0 - t_start, t1 - t0, float(ghz.mean()), float(ghz.min()), float(ghz.max())))
stop = True
th.join()
print(f"device: {torch.cuda.get_device_name(0)}, {ncu} CUs; probe: {ITERS} iterations x 4 MFMAs per wave, one wave per SIMD; wall counter {khz.value} kHz")
print("launch  start_s  wall_ms  shader GHz under MFMA load: mean  min  max over CUs   -> dense bf16 rate at that clock (TFLOP/s)")
for i, (ts, dt, m, lo, hi) in enumerate(rows):
    print(f"{i:6d}  {ts:7.3f}  {dt * 1e3:7.1f}  {m:.3f}  {lo:.3f}  {hi:.3f}   {ncu * 4 * 32768 / 32 * m * 1e-3:.0f}")
print("rocm-smi samples (seconds relative to the first launch):")
for t, keep in samples:
    print(f"  t={t - t_start:+.2f}s  " + " | ".join(keep))

"""Writes tests/golden/yin_case.npz by importing the reference's UNMODIFIED yin.py (plain numpy + scipy.ndimage, /root/reference/yin.py) in the
build container: seeded test signals (a glide, a vowel-like harmonic stack with silence, noise) and, for each, `compute_yin(...)[0]` with the
arguments `pitch_calc` passes (yin.py:170-176) and `pitch_calc(...)` itself at two confidence thresholds / smoothing sigmas.
    python tests/golden/make_audio_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
import yin  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sr = 24000
rng = np.random.default_rng(7)
t = np.arange(int(1.2 * sr)) / sr
glide = 0.6 * np.sin(2 * np.pi * np.cumsum(120 + 200 * t / t[-1]) / sr)
vowel = sum(a * np.sin(2 * np.pi * 180 * k * t) for k, a in ((1, 0.5), (2, 0.3), (3, 0.15), (5, 0.05)))
vowel[: sr // 5] = 0.0
vowel[-sr // 6:] *= 0.01
noise = 0.1 * rng.standard_normal(len(t))
mix = 0.7 * vowel + 0.05 * rng.standard_normal(len(t))
out = {"sr": np.array(sr)}
for name, sig in (("glide", glide), ("vowel", vowel), ("noise", noise), ("mix", mix), ("odd", glide[:17777])):
    sig = sig.astype(np.float32)
    out[f"{name}/sig"] = sig
    for thr in (0.15, 0.4):
        out[f"{name}/yin_{thr}"] = yin.compute_yin(sig=sig, sr=sr, w_len=1024, w_step=256, harmo_thresh=thr)[0]
    out[f"{name}/calc_0.6_0"] = yin.pitch_calc(sig, sr, w_len=1024, w_step=256, f0_min=100.0, f0_max=500.0, confidence_threshold=0.6, gaussian_smoothing_sigma=0.0)
    out[f"{name}/calc_0.85_1"] = yin.pitch_calc(sig, sr, w_len=512, w_step=128, f0_min=50.0, f0_max=900.0, confidence_threshold=0.85, gaussian_smoothing_sigma=1.0)
np.savez_compressed(os.path.join(HERE, "yin_case.npz"), **out)
print("wrote yin_case.npz:", {k: v.shape for k, v in out.items() if k.endswith("yin_0.15")})

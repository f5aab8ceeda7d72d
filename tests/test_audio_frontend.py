"""CPU suite: the librosa-free wav -> mel front-end (glow_tts_amd/audio.py, restating Audio.py:6-52).  librosa is absent from this image and
from /root/reference, so these are property checks of the restated algorithm (PARITY UNPINNED, see the module header), plus the entry-point
façade (Train.py / Inference.py / Arg_Parser.py) importing and exposing the reference's names."""
import numpy as np


def test_mel_filterbank_is_slaney_normalised_triangles():
    from glow_tts_amd import audio
    sr, n_fft, n_mels = 24000, 2048, 80
    fb = audio.mel_filterbank(sr, n_fft, n_mels, 125, 7600)
    assert fb.shape == (80, 1025) and (fb >= 0).all()
    freqs = np.linspace(0, sr / 2, 1025)
    centers = freqs[fb.argmax(1)]
    assert (np.diff(centers) > 0).all() and centers[0] > 125 and centers[-1] < 7600
    # unit area in Hz (Slaney normalisation): sum of weights x bin width ~ 1 for filters that span several bins
    area = fb.sum(1) * (freqs[1] - freqs[0])
    assert np.allclose(area[20:], 1.0, atol=0.05)
    # below 1 kHz the Slaney scale is linear: equally spaced centres
    low = centers[centers < 900]
    assert np.allclose(np.diff(low), np.diff(low).mean(), atol=freqs[1] - freqs[0] + 1e-6)
    assert np.allclose(audio._mel_to_hz(audio._hz_to_mel([60.0, 440.0, 1000.0, 5000.0])), [60.0, 440.0, 1000.0, 5000.0])


def test_mel_generate_shape_range_and_tone_position():
    from glow_tts_amd import audio
    sr, hop, win, nfreq = 24000, 256, 1024, 1025
    t = np.arange(sr) / sr
    tone = 0.5 * np.sin(2 * np.pi * 1000.0 * t)
    mel = audio.mel_generate(tone, sr, 80, nfreq, win, hop, mel_fmin=125, mel_fmax=7600, max_abs_value=4.0)
    assert mel.shape == (1 + len(tone) // hop, 80) and mel.dtype == np.float32
    assert mel.min() >= -4.0 and mel.max() <= 4.0
    fb = audio.mel_filterbank(sr, 2048, 80, 125, 7600)
    want_bin = int(fb[:, int(round(1000.0 / (sr / 2048)))].argmax())
    assert abs(int(mel[10:-10].mean(0).argmax()) - want_bin) <= 1
    silence = audio.mel_generate(np.zeros(4096), sr, 80, nfreq, win, hop)
    assert np.allclose(silence, -4.0)                       # 20 log10(1e-7) = -140 dB -> clipped at -Max_Abs_Mel (Audio.py:41-45)
    # pre-emphasis (Audio.py:51-52): y[n] = x[n] - 0.97 x[n-1]
    x = np.array([1.0, 2.0, 3.0])
    assert np.allclose(audio.preemphasis(x), [1.0, 2.0 - 0.97, 3.0 - 1.94])


def test_audio_prep_reads_trims_and_normalises(tmp_path):
    from scipy.io import wavfile
    from glow_tts_amd import audio
    sr = 24000
    x = np.zeros(sr, np.float32)
    x[6000:18000] = 0.25 * np.sin(2 * np.pi * 300 * np.arange(12000) / sr)
    p = str(tmp_path / "a.wav")
    wavfile.write(p, sr, (x * 32767).astype(np.int16))
    y = audio.audio_prep(p, sr, trim_top_db=30)
    assert abs(np.abs(y).max() - 1.0) < 1e-6 and 11000 <= len(y) <= 13500
    import pytest
    with pytest.raises(ValueError):
        audio.audio_prep(p, 22050, allow_resample=False)


def test_entry_point_facade_exposes_the_reference_names():
    """north_star: `Train.py` / `Inference.py` entry points preserved (Train.py:49-598, Inference.py:111-313)."""
    import Arg_Parser
    import Inference
    import Train
    for name in ("Datset_Generate", "Model_Generate", "Train_Step", "Train_Epoch", "Evaluation_Step", "Evaluation_Epoch", "Load_Checkpoint",
                 "Save_Checkpoint", "Train"):
        assert callable(getattr(Train.Trainer, name)), name
    for name in ("Model_Generate", "Inference_Step", "Inference", "Load_Checkpoint"):
        assert callable(getattr(Inference.Inferencer, name)), name
    ns = Arg_Parser.Recursive_Parse({"A": {"B": 1}, "C": [1, 2]})
    assert ns.A.B == 1 and ns.C == [1, 2]


def test_reference_schema_yaml_loads_from_cwd(tmp_path, monkeypatch):
    """B3: every reference module parses ./Hyper_Parameters.yaml from the CWD at import (Modules.py:9-13); get_hp() does the same, and the
    reference's schema (every key of Hyper_Parameters.yaml:1-137, values changed) drops in without this package's extra key."""
    import yaml
    from glow_tts_amd import hparams
    d = hparams.load_yaml(hparams.DEFAULT_YAML)
    d.pop("HIP_Precision", None)
    d["Mode"], d["Decoder"]["Stack"], d["Train"]["Batch_Size"] = "SE", 7, 48
    (tmp_path / "Hyper_Parameters.yaml").write_text(yaml.dump(d))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(hparams, "_hp", None)
    hp = hparams.get_hp()
    assert hp.Mode == "SE" and hp.Decoder.Stack == 7 and hp.Train.Batch_Size == 48 and hp.Encoder.Transformer.Attention.Window_Size == 4
    assert not hasattr(hp, "HIP_Precision")
    from glow_tts_amd.modules import GlowTTS
    m = GlowTTS()                                           # no-argument constructor reading the global hp (Modules.py:17)
    assert len(m.layer_Dict["Decoder"].layer_Dict["Flows"]) == 7 and "LUT" in m.layer_Dict
    monkeypatch.setattr(hparams, "_hp", None)


def test_stft_matches_an_independent_scipy_computation():
    """`audio.stft_magnitude` restates librosa.stft (centred, reflect padding, periodic Hann of win_length zero-padded to n_fft).  Independent
    check: scipy.signal.stft on the signal padded by hand, with its 1 / sum(window) scaling undone.  (parity with librosa itself: UNPINNED.)"""
    from scipy import signal
    from glow_tts_amd import audio
    rng = np.random.default_rng(3)
    y = rng.standard_normal(9000)
    n_fft, hop, win = 2048, 256, 1024
    got = audio.stft_magnitude(y, n_fft, hop, win)
    w = np.zeros(n_fft)
    w[(n_fft - win) // 2:(n_fft - win) // 2 + win] = signal.get_window("hann", win, fftbins=True)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    _, _, Z = signal.stft(yp, window=w, nperseg=n_fft, noverlap=n_fft - hop, boundary=None, padded=False, return_onesided=True)
    want = np.abs(Z) * w.sum()
    assert got.shape == want.shape == (1025, 1 + len(y) // hop)
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9)


def test_mel_filterbank_known_answers_and_independent_triangles():
    """Two values librosa's documentation prints for `librosa.filters.mel(22050, 2048)` (128 filters, Slaney scale and normalisation): weight
    [0, 1] = 0.016, and 0.02 with fmax = 8000; the Slaney scale's fixed points; and every filter rebuilt by an explicit per-filter loop."""
    from glow_tts_amd import audio
    assert round(float(audio.mel_filterbank(22050, 2048, 128, 0.0, 11025.0)[0, 1]), 3) == 0.016
    assert round(float(audio.mel_filterbank(22050, 2048, 128, 0.0, 8000.0)[0, 1]), 2) == 0.02
    assert np.allclose(audio._hz_to_mel([0.0, 200.0 / 3, 1000.0]), [0.0, 1.0, 15.0]) and np.isclose(audio._mel_to_hz(15.0 + 27.0), 6400.0)
    sr, n_fft, n_mels, fmin, fmax = 24000, 2048, 80, 125.0, 7600.0
    fb = audio.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    edges = audio._mel_to_hz(np.linspace(audio._hz_to_mel(fmin), audio._hz_to_mel(fmax), n_mels + 2))
    freqs = np.arange(1 + n_fft // 2) * sr / n_fft
    for m in range(n_mels):
        lo, ce, hi = edges[m], edges[m + 1], edges[m + 2]
        tri = np.where(freqs <= ce, (freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)).clip(min=0.0)
        assert np.allclose(fb[m], tri * 2.0 / (hi - lo), atol=1e-7), m


def test_yin_pitch_matches_the_reference_yin_py_vectors():
    """GR-mode pitch track (Pattern_Generator.py:41-52 -> yin.py:159-183).  tests/golden/yin_case.npz was written by the reference's own yin.py
    (tests/golden/make_audio_golden.py): `compute_yin(...)[0]` at two thresholds and `pitch_calc` - whose window / hop / f0 arguments the
    reference ignores (always 1024 / 256, 100..500 Hz) - on a glide, a harmonic stack with silence, noise, a mixture and an odd-length signal."""
    import os
    from glow_tts_amd import audio
    from glow_tts_amd.hparams import Recursive_Parse
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "yin_case.npz"))
    sr = int(d["sr"])
    voiced = 0
    for name in ("glide", "vowel", "noise", "mix", "odd"):
        sig = d[f"{name}/sig"]
        for thr in (0.15, 0.4):
            want = d[f"{name}/yin_{thr}"]
            got = audio.yin_pitch(sig, sr, harmo_thresh=thr)
            assert got.shape == want.shape and np.allclose(got, want, rtol=0, atol=1e-9), (name, thr, np.abs(got - want).max())
            voiced += int((want > 0).sum())
        for key, conf, sigma in (("calc_0.6_0", 0.6, 0.0), ("calc_0.85_1", 0.85, 1.0)):
            hp = Recursive_Parse({"Sound": {"Sample_Rate": sr, "Confidence_Threshold": conf, "Gaussian_Smoothing_Sigma": sigma}})
            want = d[f"{name}/{key}"]
            norm = (want - want.min()) / (want.max() - want.min() + 1e-7)
            assert np.allclose(audio.pitch_generate(sig, hp), norm, atol=1e-6), (name, key)
    assert voiced > 200                                              # the vectors do exercise voiced frames


def test_resample_keeps_pitch_and_duration(tmp_path):
    """Stand-in for librosa.core.load's resampling (resampy is absent: parity UNPINNED): a 440 Hz tone at 22.05 kHz read through `audio_prep`
    at 24 kHz keeps its frequency and its duration."""
    from scipy.io import wavfile
    from glow_tts_amd import audio
    sr_in, sr_out = 22050, 24000
    x = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(sr_in) / sr_in)
    y = audio.resample(x, sr_in, sr_out)
    assert abs(len(y) - sr_out) <= 1
    spec = np.abs(np.fft.rfft(y * np.hanning(len(y))))
    assert abs(np.argmax(spec) * sr_out / len(y) - 440.0) < 2.0
    p = str(tmp_path / "t.wav")
    wavfile.write(p, sr_in, (x * 32767).astype(np.int16))
    z = audio.audio_prep(p, sr_out, trim_top_db=60)
    assert abs(len(z) - sr_out) < 600 and abs(np.abs(z).max() - 1.0) < 1e-6

"""wav -> mel front-end of the reference (`Audio.py:6-52`, used by `Pattern_Generator.Pattern_Generate` :63-79 and `Inference.py:56-58`), restated
WITHOUT librosa (this image has no librosa): numpy / scipy only.  Host code, off the hot path: it feeds reference mels to the prosody
encoder at inference time.

PARITY UNPINNED: the arithmetic of `Audio.py` lives in a third-party dependency that is absent from /root/reference and from this image
(librosa; the reference's requirements pin no version, its 2020 era is librosa 0.7 / 0.8).  What is restated here is librosa's published
algorithm for exactly the calls `Audio.py` makes - `stft` (centered, reflect padding, periodic Hann window of `win_length` zero-padded to
`n_fft`), `filters.mel` (Slaney scale, Slaney area normalisation), `effects.trim` (frame RMS against the peak, `top_db`), `util.normalize` (peak) -
and checked by properties in tests/test_audio_frontend.py, not against librosa outputs.  `librosa.core.load`'s resampling is NOT restated: the
wav must already be at hp.Sound.Sample_Rate.  The YIN pitch tracker (`yin.py`, GR mode only) is out of scope."""
import numpy as np
from scipy import signal
from scipy.io import wavfile


def preemphasis(audio, pre_emphasis=0.97):
    """Audio.py:51-52."""
    return signal.lfilter([1.0, -pre_emphasis], [1.0], audio)


def _frames(y, frame_length, hop_length):
    n = 1 + (len(y) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    return y[idx]


def stft_magnitude(y, n_fft, hop_length, win_length):
    """|librosa.stft(y, n_fft, hop_length, win_length)| -> [1 + n_fft / 2, frames]; center = True, reflect padding, Hann window."""
    win = signal.get_window("hann", win_length, fftbins=True)
    lpad = (n_fft - win_length) // 2
    win = np.pad(win, (lpad, n_fft - win_length - lpad))
    y = np.pad(np.asarray(y, dtype=np.float64), n_fft // 2, mode="reflect")
    return np.abs(np.fft.rfft(_frames(y, n_fft, hop_length) * win[None, :], axis=1)).T


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax): triangular filters on the Slaney mel scale, each normalised to unit area in Hz."""
    fft_freqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_freqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0, np.minimum(lower, upper))
    weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return weights.astype(np.float32)


def mel_generate(audio, sample_rate, num_mel, num_frequency, window_length, hop_length, pre_emphasis=0.97, mel_fmin=125, mel_fmax=7600,
                 min_level_db=-100, max_abs_value=4.0):
    """Audio.py:14-47 `Mel_Generate` -> [frames, num_mel] in [-max_abs_value, max_abs_value]."""
    n_fft = (num_frequency - 1) * 2
    mag = stft_magnitude(preemphasis(audio, pre_emphasis), n_fft, hop_length, window_length)
    mag = mel_filterbank(sample_rate, n_fft, num_mel, mel_fmin, mel_fmax) @ mag
    db = 20 * np.log10(mag + 1e-7)
    return np.clip((2 * max_abs_value) * (db - min_level_db) / -min_level_db - max_abs_value, -max_abs_value, max_abs_value).T.astype(np.float32)


def trim(audio, top_db=60, frame_length=512, hop_length=256):
    """librosa.effects.trim: drop leading / trailing frames whose RMS is more than top_db below the loudest frame."""
    y = np.pad(audio, frame_length // 2, mode="reflect")
    rms = np.sqrt(np.mean(_frames(y, frame_length, hop_length) ** 2, axis=1))
    db = 20 * np.log10(np.maximum(rms, 1e-10) / max(rms.max(), 1e-10))
    keep = np.flatnonzero(db > -top_db)
    if keep.size == 0:
        return audio[:0]
    return audio[keep[0] * hop_length:min(len(audio), (keep[-1] + 1) * hop_length)]


def audio_prep(path, sample_rate, trim_top_db=60):
    """Audio.py:6-11 `Audio_Prep` (no resampling: the file's rate must equal sample_rate)."""
    sr, x = wavfile.read(path)
    if sr != sample_rate:
        raise ValueError(f"{path}: {sr} Hz, expected {sample_rate} Hz (resampling is not part of this front-end)")
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim > 1:
        x = x.mean(axis=1)
    x = trim(x, trim_top_db)
    return x / max(np.abs(x).max(), 1e-10)                                      # librosa.util.normalize (peak)


def pattern_from_wav(path, hp, top_db=30):
    """`Pattern_Generate` (Pattern_Generator.py:63-79) without the pitch: -> (mel [T, Mel], None).  top_db = 30 as Inference.py:57 passes."""
    s = hp.Sound
    audio = audio_prep(path, s.Sample_Rate, top_db)
    mel = mel_generate(audio, s.Sample_Rate, s.Mel_Dim, s.Spectrogram_Dim, s.Frame_Length, s.Frame_Shift, mel_fmin=s.Mel_F_Min, mel_fmax=s.Mel_F_Max,
                       max_abs_value=s.Max_Abs_Mel)
    return mel, None

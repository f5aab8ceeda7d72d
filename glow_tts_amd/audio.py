"""wav -> mel front-end of the reference (`Audio.py:6-52`, used by `Pattern_Generator.Pattern_Generate` :63-79 and `Inference.py:56-58`), restated
WITHOUT librosa (this image has no librosa): numpy / scipy only.  Host code, off the hot path: it feeds reference mels to the prosody
encoder at inference time.

PARITY UNPINNED: the arithmetic of `Audio.py` lives in a third-party dependency that is absent from /root/reference and from this image
(librosa; the reference's requirements pin no version, its 2020 era is librosa 0.7 / 0.8).  What is restated here is librosa's published
algorithm for exactly the calls `Audio.py` makes - `stft` (centered, reflect padding, periodic Hann window of `win_length` zero-padded to
`n_fft`), `filters.mel` (Slaney scale, Slaney area normalisation), `effects.trim` (frame RMS against the peak, `top_db`), `util.normalize` (peak) -
and checked in tests/test_audio_frontend.py against an INDEPENDENT computation of the same published definitions (scipy.signal.stft on the
padded signal; a per-filter loop over the Slaney triangles; the two values librosa's own documentation prints for `filters.mel(22050, 2048)`),
not against librosa outputs.  `librosa.core.load` resamples with resampy's `kaiser_best` filter, which is absent too: `resample` here is a
polyphase Kaiser resampler (scipy.signal.resample_poly) - same purpose, different filter, parity unpinned.
The YIN pitch tracker of GR mode (`yin.py`, `Pattern_Generator.py:41-52`) IS pinned: `yin.py` is plain numpy in the reference mount, and
tests/golden/make_audio_golden.py wrote `yin_case.npz` by importing it unmodified."""
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


def resample(x, sr_in, sr_out):
    """Stand-in for the resampling inside `librosa.core.load(path, sr=...)` (Audio.py:7): polyphase resampling with a Kaiser-windowed low-pass."""
    if sr_in == sr_out:
        return x
    from math import gcd
    g = gcd(int(sr_in), int(sr_out))
    return signal.resample_poly(x, int(sr_out) // g, int(sr_in) // g, window=("kaiser", 12.0)).astype(np.float32)


def audio_prep(path, sample_rate, trim_top_db=60, allow_resample=True):
    """Audio.py:6-11 `Audio_Prep`: load (mono, float, resampled to sample_rate), trim, peak-normalise."""
    sr, x = wavfile.read(path)
    if sr != sample_rate and not allow_resample:
        raise ValueError(f"{path}: {sr} Hz, expected {sample_rate} Hz")
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim > 1:
        x = x.mean(axis=1)
    x = resample(x, sr, sample_rate)
    x = trim(x, trim_top_db)
    return x / max(np.abs(x).max(), 1e-10)                                      # librosa.util.normalize (peak)


def yin_pitch(sig, sr, harmo_thresh, w_len=1024, w_step=256, f0_min=100, f0_max=500):
    """`yin.compute_yin(...)[0]` (yin.py:99-150) for all frames at once: centred reflect padding, per frame the difference function (:40-63, via
    FFT autocorrelation), its cumulative-mean normalisation (:66-80) and the first dip below `harmo_thresh` followed down to its local minimum
    (:83-96); 0 where unvoiced.  Frame bookkeeping as the reference's: `range(0, len - w_len, w_step)`."""
    sig = np.asarray(sig, dtype=np.float64)
    sig = np.pad(sig, (w_step + w_len - sig.shape[0] % w_step) // 2, mode="reflect")
    tau_min, tau_max = int(sr / f0_max), min(int(sr / f0_min), w_len)
    starts = np.arange(0, len(sig) - w_len, w_step)
    if starts.size == 0:
        return np.zeros(0)
    frames = sig[starts[:, None] + np.arange(w_len)[None, :]]                              # [F, w_len]
    csum = np.concatenate([np.zeros((len(starts), 1)), np.cumsum(frames * frames, axis=1)], axis=1)
    nfft = 1 << int(np.ceil(np.log2(w_len + tau_max)))
    spec = np.fft.rfft(frames, nfft, axis=1)
    acf = np.fft.irfft(spec * np.conj(spec), nfft, axis=1)[:, :tau_max]
    taus = np.arange(tau_max)
    df = csum[:, w_len - taus] + csum[:, w_len:w_len + 1] - csum[:, taus] - 2.0 * acf      # d(tau) = sum_j (x_j - x_{j+tau})^2
    cm = np.ones_like(df)
    cm[:, 1:] = df[:, 1:] * np.arange(1, tau_max)[None, :] / (np.cumsum(df[:, 1:], axis=1) + 1e-8)
    pitches = np.zeros(len(starts))
    for i, c in enumerate(cm):                                                               # first dip below the threshold, walked to its minimum
        below = np.flatnonzero(c[tau_min:tau_max] < harmo_thresh)
        if below.size:
            tau = tau_min + int(below[0])
            while tau + 1 < tau_max and c[tau + 1] < c[tau]:
                tau += 1
            pitches[i] = sr / tau
    return pitches


def pitch_generate(audio, hp):
    """`Pattern_Generator.Pitch_Generate` (:41-52) -> [frames] in [0, 1].  NOTE the reference's `pitch_calc` (yin.py:159-183) ignores the window /
    hop / f0 range it is handed and always analyses 1024-sample windows every 256 samples between 100 and 500 Hz; only the confidence threshold and
    the smoothing sigma of Hyper_Parameters.yaml take effect.  Reproduced as is."""
    s = hp.Sound
    pitch = yin_pitch(audio, s.Sample_Rate, harmo_thresh=1.0 - float(s.Confidence_Threshold))
    if pitch.size == 0:                                   # a clip shorter than one analysis window after trimming: no frames, no track
        return np.zeros(0, dtype=np.float32)
    if float(s.Gaussian_Smoothing_Sigma) > 0.0:
        from scipy.ndimage import gaussian_filter1d
        pitch = gaussian_filter1d(pitch, sigma=float(s.Gaussian_Smoothing_Sigma))
    return ((pitch - pitch.min()) / (pitch.max() - pitch.min() + 1e-7)).astype(np.float32)


def pattern_from_wav(path, hp, top_db=30, with_pitch=False):
    """`Pattern_Generate` (Pattern_Generator.py:54-70): -> (mel [T, Mel], pitch [T] or None).  top_db = 30 as Inference.py:57 passes."""
    s = hp.Sound
    audio = audio_prep(path, s.Sample_Rate, top_db)
    mel = mel_generate(audio, s.Sample_Rate, s.Mel_Dim, s.Spectrogram_Dim, s.Frame_Length, s.Frame_Shift, mel_fmin=s.Mel_F_Min, mel_fmax=s.Mel_F_Max,
                       max_abs_value=s.Max_Abs_Mel)
    return mel, (pitch_generate(audio, hp) if with_pitch else None)

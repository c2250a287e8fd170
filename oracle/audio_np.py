"""TEST INFRASTRUCTURE -- NumPy restatement of the reference's wav -> mel front-end and spectrogram -> wav export
(SURVEY rows N2 / N4).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED: the reference computes these with librosa (Requirements.txt:1, ``librosa>=0.7.2``), which is not
installed here and not installable (no network).  Every function below restates the librosa 0.7.2 routine the
reference calls (the oldest version its Requirements allow and the API generation its call sites need: positional
``librosa.filters.mel(sr, n_fft, ...)`` and ``np.complex`` in Audio.py:56 rule out librosa >= 0.10 / NumPy >= 1.24),
citing the reference call site.  Independent cross-checks that ARE possible here are in tests/test_audio.py: the STFT
against scipy.signal.stft, the pre-emphasis pair against its algebraic inverse, the mel basis against the Slaney
construction properties (triangles, area normalisation), Griffin-Lim self-consistency.
"""
import numpy as np
from scipy import signal


# ----------------------------------------------------------------------------------------- filters (Audio.py:11-15)
def preemphasis(x, coef=0.97):
    """reference Audio.py:11-12: signal.lfilter([1, -0.97], [1], x) (float64 result)."""
    return signal.lfilter([1, -coef], [1], x)


def inv_preemphasis(x, coef=0.97):
    """reference Audio.py:14-15: signal.lfilter([1], [1, -0.97], x)."""
    return signal.lfilter([1], [1, -coef], x)


# ----------------------------------------------------------------------------------------- librosa.core.load
# librosa.core.load(path, sr) resamples a file at another rate with librosa.core.resample(res_type='kaiser_best'), i.e.
# resampy.resample(y, sr_orig, sr, filter='kaiser_best') followed by util.fix_length(., ceil(n * ratio)) (librosa 0.7.2
# core/audio.py).  resampy (0.2.x; an un-vendored dependency of librosa, absent here like librosa itself) is band-limited
# sinc interpolation (J. O. Smith, "Digital Audio Resampling Home Page") with a pre-computed, linearly interpolated filter
# table.  Its published 'kaiser_best' design: 64 zero crossings, 2^9 table entries per crossing, Kaiser window beta =
# 14.769656459379492, roll-off 0.9475937167399596 (resampy/filters.py sinc_window + the parameters resampy documents for
# kaiser_best).  PARITY UNPINNED like the rest of this module: restated from the published algorithm, never run against
# resampy itself.
KAISER_BEST = {"num_zeros": 64, "precision": 9, "beta": 14.769656459379492, "rolloff": 0.9475937167399596}


def kaiser_best_filter():
    """resampy.filters.sinc_window(num_zeros, precision, window=kaiser(beta), rolloff): the right half of the windowed sinc,
    float64, 64 * 512 + 1 entries.  Returns (interp_win, num_table = 2^precision)."""
    nz, bits = KAISER_BEST["num_zeros"], 2 ** KAISER_BEST["precision"]
    n = bits * nz
    sinc_win = KAISER_BEST["rolloff"] * np.sinc(KAISER_BEST["rolloff"] * np.linspace(0, nz, num=n + 1, endpoint=True))
    taper = signal.windows.kaiser(2 * n + 1, KAISER_BEST["beta"])[n:]
    return taper * sinc_win, bits


def resample_kaiser_best(x, sr_orig, sr_new, t_begin=0, t_end=None):
    """resampy.resample(x, sr_orig, sr_new, filter='kaiser_best') restated loop for loop (resampy/core.py + interpn.py
    resample_f): output samples [t_begin, t_end) of the int(len(x) * ratio) it produces.  The output array has x's dtype
    and every `y[t] += weight * x[.]` rounds to it (float32 here), which is reproduced."""
    x = np.asarray(x)
    ratio = float(sr_new) / float(sr_orig)
    n_out = int(x.shape[0] * ratio)
    t_end = n_out if t_end is None else min(t_end, n_out)
    interp_win, num_table = kaiser_best_filter()
    if ratio < 1:
        interp_win = interp_win * ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * num_table)
    nwin, n_orig = interp_win.shape[0], x.shape[0]
    y = np.zeros(t_end - t_begin, dtype=x.dtype)
    time_register = 0.0
    for t in range(t_end):
        if t >= t_begin:
            n = int(time_register)
            frac = scale * (time_register - n)
            index_frac = frac * num_table
            offset = int(index_frac)
            eta = index_frac - offset
            acc = x.dtype.type(0)
            for i in range(min(n + 1, (nwin - offset) // index_step)):           # left wing
                weight = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step]
                acc = x.dtype.type(acc + weight * x[n - i])
            frac = scale - frac
            index_frac = frac * num_table
            offset = int(index_frac)
            eta = index_frac - offset
            for k in range(min(n_orig - n - 1, (nwin - offset) // index_step)):  # right wing
                weight = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step]
                acc = x.dtype.type(acc + weight * x[n + k + 1])
            y[t - t_begin] = acc
        time_register += time_increment          # (accumulated, as resampy does: not t * increment)
    return y


def fix_length(y, n):
    """librosa.util.fix_length: crop or zero-pad to n samples."""
    return y[:n] if y.shape[0] >= n else np.pad(y, (0, n - y.shape[0]))


def load_wav(path, sample_rate, max_out=None):
    """librosa.core.load(path, sr) for PCM wav files (Pattern_Generator.py:40-43): float32 in [-1, 1) = int / 2^(bits-1),
    channel mean, and for a file at another rate resample(res_type='kaiser_best') + fix_length(ceil(n * ratio)).
    (`max_out` bounds the resampled part for tests: the restatement is a Python loop.)"""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if data.dtype == np.int16:
        y = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        y = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        y = (data.astype(np.float32) - 128.0) / 128.0
    else:
        y = data.astype(np.float32)
    if y.ndim > 1:
        y = y.mean(axis=1)
    if sr != sample_rate:
        n = int(np.ceil(y.shape[0] * float(sample_rate) / sr))
        y = resample_kaiser_best(y, sr, sample_rate, 0, max_out)
        if max_out is None:
            y = fix_length(y, n)
    return y


# ----------------------------------------------------------------------------------------- librosa.util.frame / rms / trim
def frame(y, frame_length, hop_length):
    """librosa.util.frame (0.7.2): [frame_length, n_frames] view, n_frames = 1 + (len - frame_length) // hop."""
    n_frames = 1 + (len(y) - frame_length) // hop_length
    idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return y[idx]


def rms(y, frame_length, hop_length):
    """librosa.feature.rms(y=..., center=True, pad_mode='reflect') (0.7.2 spectral.py)."""
    y = np.pad(y, int(frame_length // 2), mode="reflect")
    x = frame(y, frame_length, hop_length)
    return np.sqrt(np.mean(np.abs(x) ** 2, axis=0, keepdims=True))


def power_to_db(S, ref, amin=1e-10):
    """librosa.core.power_to_db(S, ref=np.max, top_db=None)."""
    return 10.0 * np.log10(np.maximum(amin, S)) - 10.0 * np.log10(np.maximum(amin, ref))


def trim_bounds(y, top_db, frame_length, hop_length):
    """librosa.effects.trim (0.7.2 effects.py) -> (start, end) sample indices (reference Pattern_Generator.py:45)."""
    mse = rms(y, frame_length, hop_length) ** 2
    non_silent = power_to_db(mse.squeeze(), ref=np.max(mse)) > -top_db
    nonzero = np.flatnonzero(non_silent)
    if nonzero.size > 0:
        start = int(nonzero[0] * hop_length)
        end = min(y.shape[-1], int((nonzero[-1] + 1) * hop_length))
    else:
        start, end = 0, 0
    return start, end


# ----------------------------------------------------------------------------------------- librosa.stft / istft
def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True), what librosa.filters.get_window returns."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def pad_center(w, size):
    lpad = (size - len(w)) // 2
    return np.pad(w, (lpad, size - len(w) - lpad), mode="constant")


def stft(y, n_fft, hop_length, win_length):
    """librosa.stft(y, n_fft, hop_length, win_length) with the 0.7.2 defaults window='hann', center=True,
    pad_mode='reflect', dtype=complex64 (reference Audio.py:70-72).  Returns [1 + n_fft/2, n_frames]."""
    w = pad_center(hann_periodic(win_length), n_fft).reshape(-1, 1)
    y = np.pad(y, int(n_fft // 2), mode="reflect")
    frames = frame(y, n_fft, hop_length)
    return np.fft.rfft(w * frames, axis=0).astype(np.complex64)


def window_sumsquare(n_frames, hop_length, win_length, n_fft):
    """librosa.filters.window_sumsquare(window='hann', norm=None, dtype=float32)."""
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=np.float32)
    win_sq = pad_center(hann_periodic(win_length) ** 2, n_fft)
    for i in range(n_frames):
        s = i * hop_length
        x[s:min(n, s + n_fft)] += win_sq[:max(0, min(n_fft, n - s))]
    return x


def istft(D, hop_length, win_length):
    """librosa.istft(D, hop_length, win_length) (0.7.2: window='hann', center=True, dtype=float32, length=None)
    (reference Audio.py:74-75)."""
    n_fft = 2 * (D.shape[0] - 1)
    w = pad_center(hann_periodic(win_length), n_fft)[:, None]
    n_frames = D.shape[1]
    expected = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(expected, dtype=np.float32)
    ytmp = w * np.fft.irfft(D, axis=0)
    for i in range(n_frames):                                   # overlap-add
        y[i * hop_length:i * hop_length + n_fft] += ytmp[:, i]
    wss = window_sumsquare(n_frames, hop_length, win_length, n_fft)
    nz = wss > np.finfo(np.float32).tiny                        # librosa.util.tiny(float32)
    y[nz] /= wss[nz]
    return y[int(n_fft // 2):-int(n_fft // 2)]                  # center=True, length=None


# ----------------------------------------------------------------------------------------- librosa.filters.mel
def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sample_rate, n_fft, n_mels):
    """librosa.filters.mel(sr, n_fft, n_mels) (0.7.2: fmin=0, fmax=sr/2, htk=False, norm=1 'Slaney', float32)
    (reference Audio.py:81-83).  [n_mels, 1 + n_fft/2]."""
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sample_rate) / 2, 1 + n_fft // 2, endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(float(sample_rate) / 2), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


# ----------------------------------------------------------------------------------------- Audio.py:29-32, 49-55, 86-102
def amp_to_db(x):
    return 20 * np.log10(np.maximum(1e-5, x))                                                   # Audio.py:86-87


def symmetric_normalize(S, min_level_db=-100, max_abs_value=4):
    return np.clip((2 * max_abs_value) * ((S - min_level_db) / (-min_level_db)) - max_abs_value,
                   -max_abs_value, max_abs_value)                                                # Audio.py:95-96


def normalize(S, min_level_db=-100):
    return np.clip((S - min_level_db) / -min_level_db, 0, 1)                                    # Audio.py:92-93


def symmetric_denormalize(S, min_level_db=-100, max_abs_value=4):
    return ((np.clip(S, -max_abs_value, max_abs_value) + max_abs_value) / (2 * max_abs_value)
            * -min_level_db) + min_level_db                                                      # Audio.py:101-102


def denormalize(S, min_level_db=-100):
    return (np.clip(S, 0, 1) * -min_level_db) + min_level_db                                    # Audio.py:98-99


def magnitude(y, n_fft, hop_length, win_length):
    return np.abs(stft(preemphasis(y), n_fft, hop_length, win_length))                         # Audio.py:49-55


def melspectrogram(y, hp_sound):
    """reference Audio.py:29-32 -> [Mel_Dim, n_frames]."""
    n_fft = (int(hp_sound["Spectrogram_Dim"]) - 1) * 2
    M = magnitude(y, n_fft, int(hp_sound["Frame_Shift"]), int(hp_sound["Frame_Length"]))
    S = amp_to_db(np.dot(mel_basis(int(hp_sound["Sample_Rate"]), n_fft, int(hp_sound["Mel_Dim"])), M))
    mx = hp_sound.get("Max_Abs_Mel")
    return normalize(S) if mx is None else symmetric_normalize(S, max_abs_value=mx)


def mel_generate(sig, hp_sound, top_db=60):
    """reference Pattern_Generator.py:39-60 after the load: pre-emphasis, trim (frame 32 / hop 16) x 0.99,
    inverse pre-emphasis, melspectrogram -> [n_frames, Mel_Dim] float32."""
    sig = preemphasis(np.asarray(sig))
    start, end = trim_bounds(sig, top_db, 32, 16)
    sig = sig[start:end] * 0.99
    sig = inv_preemphasis(sig)
    return np.transpose(melspectrogram(sig, hp_sound).astype(np.float32))


# ----------------------------------------------------------------------------------------- Audio.py:23-27, 57-68 (N4)
def griffin_lim(S, hop_length, win_length, iters=60, angles0=None, rng=None):
    """reference Audio.py:57-68.  ``angles0`` (uniform [0,1) phases / 2 pi) can be injected; the reference draws
    them unseeded with np.random.rand."""
    if angles0 is None:
        angles0 = (rng or np.random).random(S.shape) if rng is not None else np.random.rand(*S.shape)
    angles = np.exp(2j * np.pi * angles0)
    S_complex = np.abs(S).astype(np.complex128)
    n_fft = 2 * (S.shape[0] - 1)
    y = istft(S_complex * angles, hop_length, win_length)
    for _ in range(iters):
        angles = np.exp(1j * np.angle(stft(y, n_fft, hop_length, win_length)))
        y = istft(S_complex * angles, hop_length, win_length)
    return y


def inv_spectrogram(spec, hp_sound, ref_level_db=20, power=1.5, max_abs_value=None, iters=60, angles0=None):
    """reference Audio.py:23-27; ``spec`` is [Spectrogram_Dim, n_frames]."""
    spec = denormalize(spec) if max_abs_value is None else symmetric_denormalize(spec, max_abs_value=max_abs_value)
    S = np.power(10.0, (spec + ref_level_db) * 0.05)
    return inv_preemphasis(griffin_lim(S ** power, int(hp_sound["Frame_Shift"]), int(hp_sound["Frame_Length"]),
                                       iters, angles0))

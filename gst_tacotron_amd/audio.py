"""Host side of the audio front end: reading reference wavs.

The DSP itself (pre-emphasis, trim, STFT, mel, dB, normalisation -- reference Pattern_Generator.py:39-60,
Audio.py:29-32) runs on the GPU behind ``gsttaco_mel_frontend``; this module only does what
``librosa.core.load(path, sr)`` does for the reference (Pattern_Generator.py:40-43): decode a PCM wav to
float32 in [-1, 1), average the channels, bring it to ``Sound.Sample_Rate``.
"""
import wave

import numpy as np


# resampy's published 'kaiser_best' design (the filter librosa.core.load resamples with): 64 zero crossings, 2^9 table entries
# per crossing, Kaiser beta, roll-off
_KB_ZEROS, _KB_BITS, _KB_BETA, _KB_ROLLOFF = 64, 512, 14.769656459379492, 0.9475937167399596
_kb_cache = {}


def _kaiser_best():
    if "w" not in _kb_cache:
        from scipy.signal.windows import kaiser
        n = _KB_BITS * _KB_ZEROS
        sinc = _KB_ROLLOFF * np.sinc(_KB_ROLLOFF * np.linspace(0, _KB_ZEROS, num=n + 1, endpoint=True))
        _kb_cache["w"] = kaiser(2 * n + 1, _KB_BETA)[n:] * sinc
    return _kb_cache["w"]


def resample_kaiser_best(x, sr_orig, sr_new, chunk=8192):
    """What ``librosa.core.load`` does to a file at another rate (Pattern_Generator.py:40-43 on e.g. the 22.05 kHz LJ wav of
    Inference_Wav_for_Training.txt:8): ``resampy.resample(x, sr_orig, sr_new, filter='kaiser_best')`` -- band-limited sinc
    interpolation with a linearly interpolated table of the windowed sinc, both wings summed per output sample -- then
    ``fix_length`` to ceil(n * ratio) samples.  Same table, same tap indices and interpolation weights as resampy's loop;
    vectorised over output samples and summed in float64 (resampy rounds to float32 after every tap: 1e-7 apart;
    oracle/audio_np.py reproduces that rounding and tests/test_audio.py compares the two)."""
    x = np.asarray(x, dtype=np.float32)
    ratio = float(sr_new) / float(sr_orig)
    n_orig, n_out = x.shape[0], int(x.shape[0] * ratio)
    win = _kaiser_best() * (ratio if ratio < 1 else 1.0)
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    step = int(scale * _KB_BITS)
    nwin = win.shape[0]
    # resampy accumulates its time register (time += 1 / ratio per output sample): reproduce the accumulated values
    time = np.concatenate([[0.0], np.cumsum(np.full(max(n_out - 1, 0), 1.0 / ratio))])[:n_out]
    y = np.zeros(n_out, dtype=np.float64)
    x64 = x.astype(np.float64)
    taps = np.arange((nwin + step - 1) // step)
    for t0 in range(0, n_out, chunk):
        tr = time[t0:t0 + chunk]
        n = tr.astype(np.int64)
        for wing in (0, 1):
            frac = scale * (tr - n) if wing == 0 else scale - scale * (tr - n)
            index_frac = frac * _KB_BITS
            offset = index_frac.astype(np.int64)
            eta = index_frac - offset
            count = np.minimum((n + 1) if wing == 0 else (n_orig - n - 1), (nwin - offset) // step)
            idx = offset[:, None] + taps[None, :] * step
            ok = taps[None, :] < count[:, None]
            idx = np.where(ok, idx, 0)
            w = win[idx] + eta[:, None] * delta[idx]
            src = (n[:, None] - taps[None, :]) if wing == 0 else (n[:, None] + taps[None, :] + 1)
            y[t0:t0 + chunk] += np.where(ok, w * x64[np.clip(src, 0, n_orig - 1)], 0.0).sum(axis=1)
    y = y.astype(np.float32)
    n_fix = int(np.ceil(n_orig * ratio))                  # librosa.util.fix_length
    return y[:n_fix] if y.shape[0] >= n_fix else np.pad(y, (0, n_fix - y.shape[0]))


def load_wav(path, sample_rate):
    """float32 mono samples at ``sample_rate``: what ``librosa.core.load(path, sr)`` returns for a PCM wav (8/16/32-bit via the
    standard library): int / 2^(bits-1), channel mean, and for a file at another rate the 'kaiser_best' resampling above."""
    with wave.open(path, "rb") as f:
        nch, width, sr, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        y = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        y = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        y = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError("{}: unsupported PCM sample width {}".format(path, width))
    if nch > 1:
        y = y.reshape(-1, nch).mean(axis=1).astype(np.float32)
    if sr != sample_rate:
        y = resample_kaiser_best(y, sr, sample_rate)
    return y


def as_signal(item, sample_rate):
    """A path -> samples; a 1-D array is taken as samples already at ``sample_rate``."""
    if isinstance(item, (str, bytes)) or hasattr(item, "__fspath__"):
        return load_wav(item, sample_rate)
    a = np.asarray(item, dtype=np.float32)
    if a.ndim != 1:
        raise ValueError("a reference signal must be a wav path or a 1-D sample array")
    return a

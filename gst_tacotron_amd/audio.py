"""Host side of the audio front end: reading reference wavs.

The DSP itself (pre-emphasis, trim, STFT, mel, dB, normalisation -- reference Pattern_Generator.py:39-60,
Audio.py:29-32) runs on the GPU behind ``gsttaco_mel_frontend``; this module only does what
``librosa.core.load(path, sr)`` does for the reference (Pattern_Generator.py:40-43): decode a PCM wav to
float32 in [-1, 1), average the channels, bring it to ``Sound.Sample_Rate``.
"""
import wave

import numpy as np


def load_wav(path, sample_rate):
    """float32 mono samples at ``sample_rate``.  8/16/32-bit PCM via the standard library.

    Deviation from the reference: librosa resamples with resampy 'kaiser_best'; here a file at another rate goes
    through ``scipy.signal.resample_poly`` (polyphase FIR) -- close, not sample-identical.  Files already at
    ``Sound.Sample_Rate`` (all FastVox reference wavs of the reference repo) are untouched."""
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
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sample_rate), int(sr))
        y = resample_poly(y.astype(np.float64), int(sample_rate) // g, int(sr) // g).astype(np.float32)
    return y


def as_signal(item, sample_rate):
    """A path -> samples; a 1-D array is taken as samples already at ``sample_rate``."""
    if isinstance(item, (str, bytes)) or hasattr(item, "__fspath__"):
        return load_wav(item, sample_rate)
    a = np.asarray(item, dtype=np.float32)
    if a.ndim != 1:
        raise ValueError("a reference signal must be a wav path or a 1-D sample array")
    return a

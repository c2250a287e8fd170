"""Writes tests/golden/audio_lj_excerpt.npz: the first 1.5 s (int16 PCM, 22.05 kHz) of the reference repo's one reference wav
that is NOT at Sound.Sample_Rate -- /root/reference/Wav_for_Inference/LJ.LJ050-0278.wav, Inference_Wav_for_Training.txt:8 --
and what the oracle's restatement of librosa.core.load's 'kaiser_best' resampling (oracle/audio_np.py resample_kaiser_best,
resampy's loop with its float32 rounding) makes of it at 16 kHz.  Run in the build container (needs /root/reference):

    python -m oracle.gen_golden_audio_lj

TEST INFRASTRUCTURE.  Data only.  The expected samples are ORACLE outputs (resampy / librosa are not installable here: parity
unpinned for the audio path, see oracle/audio_np.py).
"""
import os

import numpy as np

from oracle import audio_np as A

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "audio_lj_excerpt.npz")
REF_WAV = "/root/reference/Wav_for_Inference/LJ.LJ050-0278.wav"


def main():
    from scipy.io import wavfile
    sr, pcm = wavfile.read(REF_WAV)
    assert sr == 22050 and pcm.dtype == np.int16 and pcm.ndim == 1
    pcm = pcm[:33075]
    y = A.resample_kaiser_best(pcm.astype(np.float32) / 32768.0, sr, 16000)
    y = A.fix_length(y, int(np.ceil(pcm.shape[0] * 16000.0 / sr)))
    np.savez_compressed(OUT, pcm=pcm, sample_rate=np.array(sr), target_rate=np.array(16000), resampled=y)
    print(OUT, os.path.getsize(OUT), "bytes;", pcm.shape[0], "->", y.shape[0], "samples, peak", float(np.abs(y).max()))


if __name__ == "__main__":
    main()

"""Writes tests/golden/audio_*.npz: inputs + oracle outputs of the wav -> mel front end and of
Griffin-Lim (N4).  Run in the build container (needs /root/reference for the FastVox reference wav):

    python -m oracle.gen_golden_audio

TEST INFRASTRUCTURE.  The expected values are ORACLE outputs (librosa is not installable here: parity unpinned, see
oracle/audio_np.py).  The one real-speech input is the smallest reference wav of the reference repo
(Wav_for_Inference/FV.KSP.arctic_a0005.wav, CMU Arctic, 16 kHz int16), stored as its int16 samples.
"""
import os

import numpy as np

from gst_tacotron_amd import hparams, synthetic
from oracle import audio_np as A

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
REF_WAV = "/root/reference/Wav_for_Inference/FV.KSP.arctic_a0005.wav"


def synthetic_signals(seed, sr, n):
    """Ragged batch: silence, a noise burst with an envelope, a chirp, trailing low-level noise."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        lead, body, tail = rng.integers(200, 3000), rng.integers(sr // 4, sr), rng.integers(100, 4000)
        t = np.arange(body) / sr
        env = np.sin(np.pi * np.arange(body) / body) ** 2
        chirp = 0.4 * np.sin(2 * np.pi * (200 + 1500 * t) * t)
        sig = env * (chirp + 0.05 * rng.standard_normal(body))
        y = np.concatenate([1e-4 * rng.standard_normal(lead), sig, 3e-4 * rng.standard_normal(tail)])
        out.append(y.astype(np.float32))
    return out


def main():
    from scipy.io import wavfile
    hp = hparams.load_hp()["Sound"]
    sr, pcm = wavfile.read(REF_WAV)
    assert sr == hp["Sample_Rate"] and pcm.dtype == np.int16
    y = pcm.astype(np.float32) / 32768.0
    pre = A.preemphasis(y)
    out = {"sample_rate": np.array(sr), "pcm": pcm}
    for top_db in (60, 15):
        out["bounds_%d" % top_db] = np.array(A.trim_bounds(pre, top_db, 32, 16))
        out["mel_%d" % top_db] = A.mel_generate(y, hp, top_db)
    np.savez_compressed(os.path.join(OUT, "audio_fv_ksp.npz"), **out)
    print("audio_fv_ksp", out["mel_60"].shape, out["mel_15"].shape, out["bounds_60"], out["bounds_15"])

    # tiny Sound section (n_fft 64, hop 16, 16 mel bands) and the reference's, on synthetic ragged signals
    for name, snd, seed in (("tiny", synthetic.tiny_hp()["Sound"], 11), ("full", hp, 12)):
        sigs = synthetic_signals(seed, snd["Sample_Rate"], 5)
        o = {"n": np.array(len(sigs)), "sound_json": np.array(__import__("json").dumps(snd))}
        for i, s in enumerate(sigs):
            o["sig%d" % i] = s
            for top_db in (60, 15):
                o["mel%d_%d" % (i, top_db)] = A.mel_generate(s, snd, top_db)
        np.savez_compressed(os.path.join(OUT, "audio_synth_%s.npz" % name), **o)
        print("audio_synth_" + name, [o["mel%d_60" % i].shape[0] for i in range(len(sigs))],
              [o["mel%d_15" % i].shape[0] for i in range(len(sigs))])


if __name__ == "__main__":
    main()

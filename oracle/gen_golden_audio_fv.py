"""Writes tests/golden/audio_fv_all.npz: the int16 PCM of the reference repo's seven FastVox (CMU Arctic) reference wavs
-- /root/reference/Wav_for_Inference/FV.*.wav, the multi-speaker references of Inference_Wav_for_Training.txt:1-7 that
BASELINE configs[4] names -- plus what the oracle's wav -> mel front end (oracle/audio_np.py, Pattern_Generator.py:39-60
restated) makes of them at top_db = 15, the value the reference's Feeder uses for several references (Feeder.py:204-209).
Run in the build container (needs /root/reference):

    python -m oracle.gen_golden_audio_fv

TEST INFRASTRUCTURE.  Data only: samples and arrays, no reference source text.  The expected mels are ORACLE outputs
(librosa is not installable here: parity unpinned for the audio path, see oracle/audio_np.py).
"""
import glob
import os

import numpy as np

from gst_tacotron_amd import hparams
from oracle import audio_np as A

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "audio_fv_all.npz")
REF_DIR = "/root/reference/Wav_for_Inference"


def main():
    from scipy.io import wavfile
    hp = hparams.load_hp()["Sound"]
    paths = sorted(glob.glob(os.path.join(REF_DIR, "FV.*.wav")))
    assert len(paths) == 7, paths
    out = {"n": np.array(len(paths)), "sample_rate": np.array(hp["Sample_Rate"]), "top_db": np.array(15)}
    for i, p in enumerate(paths):
        sr, pcm = wavfile.read(p)
        assert sr == hp["Sample_Rate"] and pcm.dtype == np.int16 and pcm.ndim == 1
        y = pcm.astype(np.float32) / 32768.0
        out["name%d" % i] = np.array(os.path.basename(p))
        out["pcm%d" % i] = pcm
        out["bounds%d" % i] = np.array(A.trim_bounds(A.preemphasis(y), 15, 32, 16))
        out["mel%d" % i] = A.mel_generate(y, hp, 15).astype(np.float32)
        print(os.path.basename(p), pcm.shape[0], "samples ->", out["mel%d" % i].shape[0], "frames, trim", out["bounds%d" % i])
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

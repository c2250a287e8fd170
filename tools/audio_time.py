"""Times the audio front end (N2) and back end (N4) at serving shapes, with the NumPy oracle beside them."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gst_tacotron_amd import hparams
from gst_tacotron_amd.model import GST_Tacotron

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=4.0)
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--cpu", action="store_true")
a = ap.parse_args()
hp = hparams.load_hp()
sr = hp["Sound"]["Sample_Rate"]
m = GST_Tacotron(hyper_parameters=hp, max_batch=a.batch, max_tokens=8, max_ref_frames=4, max_wav_seconds=a.seconds + 1)
rng = np.random.default_rng(0)
n = int(a.seconds * sr)
sigs = [(0.3 * np.sin(np.arange(n) * 0.05 * (i + 1)) * np.hanning(n) + 0.01 * rng.standard_normal(n)).astype(np.float32)
        for i in range(a.batch)]


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


ms = timed(lambda: m.Mel_Generate(sigs, 60))
mels, lens = m.Mel_Generate(sigs, 60)
print("front end: %d wavs x %.1f s -> %d frames each: %.3f ms per batch incl. H2D of %.1f MB and the length read-back"
      % (a.batch, a.seconds, int(lens[0]), ms, a.batch * n * 4 / 1e6))
spec = (torch.rand(a.batch, a.frames, hp["Sound"]["Spectrogram_Dim"], device="cuda") * 8 - 4)
ms = timed(lambda: m.Inv_Spectrogram(spec, iters=a.iters), reps=3)
audio_s = a.batch * hp["Sound"]["Frame_Shift"] * (a.frames - 1) / sr
print("Griffin-Lim: %d x %d frames, %d iterations: %.2f ms per batch = %.0fx real time (%.1f s of audio)"
      % (a.batch, a.frames, a.iters, ms, audio_s / (ms / 1e3), audio_s))
if a.cpu:
    from oracle import audio_np as A
    t = time.perf_counter(); A.mel_generate(sigs[0], hp["Sound"], 60); t1 = time.perf_counter() - t
    s1 = spec[0].cpu().numpy().T.astype(np.float64)
    t = time.perf_counter(); A.inv_spectrogram(s1, hp["Sound"], max_abs_value=4, iters=a.iters); t2 = time.perf_counter() - t
    print("NumPy oracle, ONE utterance on one core: front end %.1f ms, Griffin-Lim %.0f ms" % (t1 * 1e3, t2 * 1e3))

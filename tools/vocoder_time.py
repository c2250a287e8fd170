"""Time gsttaco_vocoder at the cfg2 output shape: B utterances x T mel frames -> 513-bin spectrograms."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
hp = synthetic.config_hp("cfg2")
m = GST_Tacotron(hyper_parameters=hp, max_batch=a.batch, max_tokens=8, max_ref_frames=2)
m.Restore(weights=weights.synthetic_weights(hp, seed=0))
mel = torch.randn(a.batch, a.frames, 80, device="cuda")
for _ in range(2):
    m.vocoder(mel)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    m.vocoder(mel)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
print("vocoder B=%d T=%d: %.3f ms -> %.0f frames/s" % (a.batch, a.frames, ms, a.batch * a.frames / ms * 1e3))

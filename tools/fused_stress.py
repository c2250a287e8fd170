"""Stress check of the in-kernel hand-offs: whole Inference_Steps (all 500 decode steps, throughput-mode randomness) with the fused
LSTM launch -- or, with --persist, the persistent decode launch (round 4) -- repeated, against the two-launch form, bitwise.  A stale
read in a hand-off would show up as a difference.
    python tools/fused_stress.py [batch] [reps] [--mixed] [--persist] [--lsa]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 10
hp, inputs = synthetic.config_inputs("cfg2", batch=B)
hp["Use_Mixed_Precision"] = "--mixed" in sys.argv
if "--lsa" in sys.argv:        # the LSA chain of the one-group kernel (round 5)
    hp["Tacotron2"]["Decoder"]["Attention"] = {"Type": "LSA", "Size": 128, "Conv": {"Filters": 32, "Kernel_Size": 31}, "Smoothing": False}
w = weights.synthetic_weights(hp, seed=0)
outs = {}
persist = "--persist" in sys.argv
for flag in ("0", "1"):
    os.environ["GSTTACO_FUSED_LSTM"] = flag
    os.environ["GSTTACO_PERSIST_DECODE"] = flag if persist else "0"
    m = GST_Tacotron(hyper_parameters=hp, max_batch=B, max_tokens=128, max_ref_frames=257)
    m.Restore(weights=w)
    res = []
    for i in range(reps):
        mel, stop, _, align = m.Inference_Step(inputs["tokens"], None, None, inputs["mels_for_gst"], inputs["mel_lengths_for_gst"], seed=100 + i)
        res.append((mel.cpu().numpy(), align.cpu().numpy()))
    m.synchronize()
    assert m.handoff_error() == 0
    assert (m.decode_counters()[0] > 0) == (persist and flag == "1")
    outs[flag] = res
    del m
    import gc
    gc.collect()
bad = sum(int(not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))) for a, b in zip(outs["0"], outs["1"]))
print("batch", B, "mixed" if hp["Use_Mixed_Precision"] else "fp32", "LSA" if "--lsa" in sys.argv else "", ":", reps, "Inference_Steps x 500 decode steps,", "persistent decode launch" if persist else "fused", "vs two launches:", bad, "differ")
sys.exit(1 if bad else 0)

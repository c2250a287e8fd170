"""How much the step-wise LSA extension loses by running on the launch path: whole Inference_Step at the headline shape
(batch 32 x 128 tokens x 1000 frames) with Attention.Type = LSA (k = 31, 32 filters) against SMA on the persistent launch and on the
launch path (GSTTACO_PERSIST_DECODE=0).    python tools/lsa_time.py"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron

def run(att, env):
    for k, v in env.items():
        os.environ[k] = v
    hp, inputs = synthetic.config_inputs("cfg2", batch=32)
    if att == "LSA":
        hp["Tacotron2"]["Decoder"]["Attention"] = {"Type": "LSA", "Size": 128, "Conv": {"Filters": 32, "Kernel_Size": 31}, "Smoothing": False}
    w = weights.synthetic_weights(hp, seed=0)
    m = GST_Tacotron(hyper_parameters=hp, max_batch=32, max_tokens=128, max_ref_frames=257)
    m.Restore(weights=w)
    args = (inputs["tokens"], None, None, inputs["mels_for_gst"], inputs["mel_lengths_for_gst"])
    for i in range(3):
        m.Inference_Step(*args, seed=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        m.Inference_Step(*args, seed=10 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print("%-4s %-28s %7.2f ms per Inference_Step, persistent decode launches %d" % (att, env or "default", ms, m.decode_counters()[0]))
    for k in env:
        del os.environ[k]
    del m
    gc.collect()        # (the persistent launch is taken only while the process has ONE live context: no lingering one)

run("SMA", {})
run("SMA", {"GSTTACO_PERSIST_DECODE": "0"})
run("SMA", {"GSTTACO_PERSIST_DECODE": "0", "GSTTACO_FUSED_FRONT": "1"})      # the general utterance kernel (LSA's), not the lean one
run("LSA", {})
run("LSA", {"GSTTACO_FUSED_FRONT": "0"})                                    # the four-kernel path (attention.hip)

"""What decoder sizes other than the reference's cost today: whole Inference_Step at the headline shape (batch 32 x 128
tokens x 1000 frames) for hyper-parameters the persistent launch does not take -- they run the launch path (3 launches per decode step).
    python tools/other_sizes_time.py"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron

def run(tag, edit, env=None):
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    hp, inputs = synthetic.config_inputs("cfg2", batch=32)
    edit(hp)
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
    plan = m.decode_plan(128)
    print("%-44s %7.2f ms per Inference_Step | persistent decode launches %d | plan (fused front, fused prenet-0, lean) %s" % (tag, ms, m.decode_counters()[0], plan))
    for k in env:
        del os.environ[k]
    del m
    gc.collect()        # (the persistent launch is taken only while the process has ONE live context: no lingering one)

dec = lambda hp: hp["Tacotron2"]["Decoder"]
run("reference sizes (256/256, 128, 1024/1024)", lambda hp: None)
run("reference sizes, launch path", lambda hp: None, {"GSTTACO_PERSIST_DECODE": "0"})
def small_prenet(hp): dec(hp)["Prenet"]["Size"] = [128, 128]
def small_lstm(hp): dec(hp)["RNN"]["Size"] = [512, 512]
def small_att(hp): dec(hp)["Attention"]["Size"] = 64
def all_small(hp): small_prenet(hp); small_lstm(hp); small_att(hp)
run("prenet 128/128", small_prenet)
run("LSTM 512/512", small_lstm)
run("attention 64", small_att)
run("prenet 128/128, LSTM 512/512, attention 64", all_small)

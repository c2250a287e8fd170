"""Throughput of one Inference_Step at other per-GPU batch sizes (BASELINE configs[2] / [4] use 128 and 64); not the headline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
for mixed in (False, True):
    for B in (16, 32, 64, 128):
        hp, inputs = synthetic.config_inputs("cfg2", batch=B)
        hp["Use_Mixed_Precision"] = mixed
        m = GST_Tacotron(hyper_parameters=hp, max_batch=B, max_tokens=128, max_ref_frames=257).Restore(weights=weights.synthetic_weights(hp, seed=0))
        dev = m.device
        tok = torch.as_tensor(inputs["tokens"]).to(dev); mels = torch.as_tensor(inputs["mels_for_gst"]).to(dev); lens = torch.as_tensor(inputs["mel_lengths_for_gst"]).to(dev)
        for i in range(2):
            m.Inference_Step(tok, None, None, mels, lens, seed=i)
        torch.cuda.synchronize()
        K = 5
        t0 = time.perf_counter()
        for i in range(K):
            m.Inference_Step(tok, None, None, mels, lens, seed=10 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print("%s batch %3d: %.2f ms per Inference_Step, %.2f M mel-frames/s" % ("bf16" if mixed else "fp32", B, 1e3 * dt, B * 1000 / dt / 1e6), flush=True)
        del m

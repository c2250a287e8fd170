"""Is the host-side cost of hipGraphLaunch (2 200 nodes) hidden behind the previous replay?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
hp, inputs = synthetic.config_inputs("cfg2", batch=32)
m = GST_Tacotron(hyper_parameters=hp, max_batch=32, max_tokens=128, max_ref_frames=257)
m.Restore(weights=weights.synthetic_weights(hp, seed=0))
dev = {k: torch.as_tensor(v).cuda() for k, v in inputs.items()}
call = lambda i: m.Inference_Step(dev["tokens"], None, None, dev["mels_for_gst"], dev["mel_lengths_for_gst"], seed=i)
for i in range(3):
    call(i)
torch.cuda.synchronize()
t = time.perf_counter(); call(0); t_call = time.perf_counter() - t; torch.cuda.synchronize(); t_one = time.perf_counter() - t
print("one replay: host returns after %.3f ms, done after %.3f ms" % (t_call * 1e3, t_one * 1e3))
K = 10
host = []
t0 = time.perf_counter()
for i in range(K):
    t = time.perf_counter(); call(i); host.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) * 1e3
print("%d back-to-back: %.3f ms per replay; host time per call:" % (K, tot / K), ["%.2f" % h for h in host])

"""Robustness check: a foreign stream keeps the GPU busy with large PyTorch GEMMs while Inference_Steps run on the persistent decode launch.
The launch needs all its workgroups co-resident; foreign workgroups can delay that.  Expected: every call either completes (the foreign
kernels drain and the launch becomes resident) or gives up within its bound, is reported by synchronize(), and the repeated call on the
launch forms is correct -- never a hang, never a silent wrong result.
    python tools/foreign_load.py [calls]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
from gst_tacotron_amd.capi import GstTacoError
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hp, inputs = synthetic.config_inputs("cfg2", batch=32)
w = weights.synthetic_weights(hp, seed=0)
m = GST_Tacotron(hyper_parameters=hp, max_batch=32, max_tokens=128, max_ref_frames=257)
m.Restore(weights=w)
args = (inputs["tokens"], None, None, inputs["mels_for_gst"], inputs["mel_lengths_for_gst"])
ref = m.Inference_Step(*args, seed=7)[0].cpu().numpy()
m.synchronize()
stop = False
side = torch.cuda.Stream()
def foreign():
    a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
    with torch.cuda.stream(side):
        while not stop:
            for _ in range(4): a @ b
            side.synchronize()
th = threading.Thread(target=foreign); th.start()
time.sleep(0.5)
ok = gave_up = 0
t0 = time.time()
for i in range(N):
    try:
        out = m.Inference_Step(*args, seed=7)[0]
        m.synchronize()
    except GstTacoError as e:
        gave_up += 1
        out = m.Inference_Step(*args, seed=7)[0]
        m.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref), "wrong result under foreign load"
    ok += 1
stop = True; th.join()
print("calls", N, "correct", ok, "give-ups", gave_up, "persistent launches", m.decode_counters(), "seconds", round(time.time() - t0, 2), "message:", m.last_message()[:80])

"""Diagnostic: phase stamps of the decode kernels at the middle decode step (GSTTACO_STAMPS=1)."""
import ctypes, os, sys
os.environ["GSTTACO_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
hp, inputs = synthetic.config_inputs("cfg2", batch=32)
if os.environ.get("STAMPS_NO_RANDOM") == "1":
    hp["Tacotron2"]["Decoder"]["Prenet"]["Dropout_Rate"] = 0.0
    hp["Tacotron2"]["Decoder"]["Attention"]["Sigmoid_Noise"] = 0.0
if os.environ.get("STAMPS_LSA") == "1":      # the LSA extension on the fused front end (dec_front_lsa.hip); launch path
    hp["Tacotron2"]["Decoder"]["Attention"] = {"Type": "LSA", "Size": 128, "Conv": {"Filters": 32, "Kernel_Size": 31}, "Smoothing": False}
w = weights.synthetic_weights(hp, seed=0)
m = GST_Tacotron(hyper_parameters=hp, max_batch=32, max_tokens=128, max_ref_frames=257)
m.Restore(weights=w)
for i in range(3):
    m.Inference_Step(inputs["tokens"], None, None, inputs["mels_for_gst"], inputs["mel_lengths_for_gst"], seed=i)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 96)()
m.ctx.check(m.ctx.lib.gsttaco_debug_stamps(m.ctx.handle, buf))
names = {0: "front", 1: "lstm1", 2: "lstm2"}
for k in range(3):
    st = [buf[k * 16 + i] for i in range(4 if k else 8)]
    st = [x for x in st if x]
    print(names[k], "phase deltas (us):", [round((b - a) / 100.0, 2) for a, b in zip(st[:-1], st[1:])], "total", round((st[-1] - st[0]) / 100.0, 2))
for k in (1, 2):
    b0 = buf[k * 16]
    print(names[k], "entry -> stamp0 (us):", (b0 - buf[k * 16 + 4]) / 100.0, " stamp0 -> loads issued:", (buf[k * 16 + 5] - b0) / 100.0)
f = [buf[i] for i in range(16)]
t0 = f[0]
print("front raw (us since stamp0):", {i: round((f[i]-t0)/100.0, 2) for i in range(16) if f[i]})
print('front prologue (us since stamp0): WG entry', round((f[12]-t0)/100.0,2), 'small loads issued', round((f[13]-t0)/100.0,2), 'seed+hash done', round((f[14]-t0)/100.0,2), 'W1 issued', round((f[15]-t0)/100.0,2))
print("front -> lstm1 start gap (us):", (buf[16] - buf[7]) / 100.0, " lstm1 end -> lstm2 start:", (buf[32] - buf[19]) / 100.0)
w = {i: round((f[i]-t0)/100.0, 2) for i in (8, 9, 10, 11) if f[i]}
print("front workers (us since utterance-WG0 stamp0): first worker start/end", w.get(8), w.get(9), " last worker start/end", w.get(10), w.get(11))
pj = [buf[40 + i] for i in range(6)]
if pj[0]:
    print("proj main WG0 (us since its entry): core done", (pj[1]-pj[0])/100.0, "after barrier", (pj[2]-pj[0])/100.0, "end", (pj[3]-pj[0])/100.0,
          "| lstm2 end -> proj entry", (pj[0]-buf[35])/100.0, "| first worker entry/end", (pj[4]-pj[0])/100.0, (pj[5]-pj[0])/100.0)

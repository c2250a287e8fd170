"""Diagnostic: phase stamps of the persistent decode launch (one-group kernel, <= 32 utterances) at the middle decode step
(GSTTACO_STAMPS=1): chain workgroup 0, projection workgroup 32, plain workgroup 255 (csrc/persist_decode.hip PD_STAMP).
    python tools/stamps_persist.py [batch <= 32] [tokens <= 256]"""
import ctypes, os, sys
os.environ["GSTTACO_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
import numpy as np
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B = nums[0] if nums else 32
Tv = nums[1] if len(nums) > 1 else 128
hp, inputs = synthetic.config_inputs("cfg2", batch=B)
if os.environ.get("STAMPS_BMA") == "1":
    hp["Tacotron2"]["Decoder"]["Attention"]["Type"] = "BMA"
if os.environ.get("STAMPS_LSA") == "1":      # the LSA chain of the one-group kernel (up to 128 tokens)
    hp["Tacotron2"]["Decoder"]["Attention"] = {"Type": "LSA", "Size": 128, "Conv": {"Filters": 32, "Kernel_Size": 31}, "Smoothing": False}
w = weights.synthetic_weights(hp, seed=0)
tok = inputs["tokens"]
if Tv != tok.shape[1]:
    tok, _ = synthetic.make_tokens(np.random.default_rng(1), B, Tv)
print("batch", B, "tokens", Tv)
m = GST_Tacotron(hyper_parameters=hp, max_batch=B, max_tokens=Tv, max_ref_frames=257)
m.Restore(weights=w)
for i in range(3):
    m.Inference_Step(tok, None, None, inputs["mels_for_gst"], inputs["mel_lengths_for_gst"], seed=i)
torch.cuda.synchronize()
assert m.decode_counters()[0] > 0, "the persistent decode launch was not taken"
buf = (ctypes.c_uint64 * 96)()
m.ctx.check(m.ctx.lib.gsttaco_debug_stamps(m.ctx.handle, buf))
names = ["chain WG 0", "proj  WG 32", "plain WG 255"]
legend = {0: "step start", 1: "prenet flags seen", 2: "prenet part multiplied", 3: "context flags seen", 4: "cell 1 done (h1 stored)", 5: "h1 arrivals seen",
          6: "cell 2 MFMAs done", 7: "h2 stored", 8: "h2 arrival counted", 9: "h2 arrivals seen (projection)", 10: "projection MFMAs done",
          11: "projection published", 12: "recurrent half 1 done", 13: "h2 arrivals seen (recurrent half)", 14: "recurrent half 2 done",
          16: "prenet published", 13 + 100: "", 17: "query done", 18: "scores done", 19: "alignment done", 20: "context published",
          23: "prenet-1 sums stored", 15: "prenet-1 barrier passed", 21: "cell 2 fragments requested", 22: "cell 2 sums reduced"}
chain_legend = dict(legend); chain_legend.update({13: "weights requested", 14: "z0 arrived, y0 in LDS"})
legend.update({21: "cell 2 fragments requested", 22: "cell 2 sums reduced"})
for r in range(3):
    t0 = buf[r * 32]
    st = sorted((buf[r * 32 + i], i) for i in range(1, 24 if r == 0 else 32) if buf[r * 32 + i] and not (r == 0 and i in (9, 10)))
    lg = chain_legend if r == 0 else legend
    print(names[r], "(us since its step start):", ", ".join("[%d %s] %.2f" % (i, lg.get(i, ""), (v - t0) / 100.0) for v, i in st))
print("step starts relative to chain WG 0's (us): proj %.2f plain %.2f" % ((buf[32] - buf[0]) / 100.0, (buf[64] - buf[0]) / 100.0))
if buf[9] and buf[24]:
    print("kernel entry of workgroup 0 -> its step 0 starts: %.1f us; workgroup 255 enters %.1f us after workgroup 0" % ((buf[24] - buf[9]) / 100.0, (buf[10] - buf[9]) / 100.0))
steps = 500
ts = [0, 1, 2, 8, 64, steps >> 1, (3 * steps) >> 2, steps - 1]
tv = [buf[24 + k] for k in range(8)]
print("chain WG 0, average step period between the stamped steps (us):",
      ", ".join("steps %d..%d: %.2f" % (ts[k], ts[k + 1], (tv[k + 1] - tv[k]) / 100.0 / (ts[k + 1] - ts[k])) for k in range(7)),
      "| first stamped step start -> last: %.1f us" % ((tv[7] - tv[0]) / 100.0))

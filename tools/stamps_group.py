"""Diagnostic: phase stamps of the GROUP kernels of the persistent decode launch (batches above 32 rows; csrc/persist_decode.hip
gt_persist_decode_g_kernel) at the middle decode step (GSTTACO_STAMPS=1): chain workgroup 0, the first projection workgroup, plain
workgroup 255.  Slot 0 = step start, 1 = chain done; per group g, slot 2 + 7 g + k: 0 context flags seen, 1 cell 1 done (h1 stored,
arrived), 2 h1 arrivals seen, 3 cell 2 done (h2 stored, arrived), 4 recurrent half 1 done (projection role: projection published),
5 h2 arrivals seen (recurrent half 2), 6 recurrent half 2 done.
    python tools/stamps_group.py [batch] [tokens] [--mixed]"""
import ctypes, os, sys
os.environ["GSTTACO_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B = nums[0] if nums else 128
Tv = nums[1] if len(nums) > 1 else 128
hp, inputs = synthetic.config_inputs("cfg2", batch=B)
hp["Use_Mixed_Precision"] = "--mixed" in sys.argv
w = weights.synthetic_weights(hp, seed=0)
tok = inputs["tokens"]
if Tv != tok.shape[1]:
    tok, _ = synthetic.make_tokens(np.random.default_rng(1), B, Tv)
m = GST_Tacotron(hyper_parameters=hp, max_batch=B, max_tokens=Tv, max_ref_frames=257)
m.Restore(weights=w)
for i in range(3):
    m.Inference_Step(tok, None, None, inputs["mels_for_gst"], inputs["mel_lengths_for_gst"], seed=i)
m.synchronize()
assert m.decode_counters()[0] > 0, "the persistent decode launch was not taken"
buf = (ctypes.c_uint64 * 96)()
m.ctx.check(m.ctx.lib.gsttaco_debug_stamps(m.ctx.handle, buf))
G = (B + 31) // 32 if B > 32 else 2
names = ["chain WG 0", "proj WG", "plain WG 255"]
ph = ["ctx flags seen", "cell1 done", "h1 seen", "cell2 done", "rec1/proj done", "h2 seen", "rec2 done"]
for r in range(3):
    t0 = buf[r * 32]
    out = []
    if r == 0 and buf[1]:
        out.append("chain done %.2f" % ((buf[1] - t0) / 100.0))
    for g in range(G):
        for k in range(7):
            v = buf[r * 32 + 2 + 7 * g + k]
            if v:
                out.append("g%d %s %.2f" % (g, ph[k], (v - t0) / 100.0))
    print(names[r], "(us since its step start):", "; ".join(out))
print("step starts relative to chain WG 0's (us): proj %.2f plain %.2f" % ((int(buf[32]) - int(buf[0])) / 100.0, (int(buf[64]) - int(buf[0])) / 100.0))

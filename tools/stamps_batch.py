"""Diagnostic: phase stamps of the decode LSTM launches at batches above 32 rows (GSTTACO_STAMPS=1; lean_body.h gt_lean_mc).
    python tools/stamps_batch.py [batch] [--mixed]"""
import ctypes, os, sys
os.environ["GSTTACO_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
hp, inputs = synthetic.config_inputs("cfg2", batch=B)
hp["Use_Mixed_Precision"] = "--mixed" in sys.argv
w = weights.synthetic_weights(hp, seed=0)
m = GST_Tacotron(hyper_parameters=hp, max_batch=B, max_tokens=128, max_ref_frames=257)
m.Restore(weights=w)
for i in range(3):
    m.Inference_Step(inputs["tokens"], None, None, inputs["mels_for_gst"], inputs["mel_lengths_for_gst"], seed=i)
torch.cuda.synchronize()
buf = (ctypes.c_uint64 * 96)()
m.ctx.check(m.ctx.lib.gsttaco_debug_stamps(m.ctx.handle, buf))
for k, name in ((1, "lstm1"), (2, "lstm2")):
    st = {i: buf[k * 16 + i] for i in range(8)}
    t0 = st[4]
    print(name, "B", B, "us since entry:", {n: round((st[i] - t0) / 100.0, 2) for i, n in
          ((0, "first loads requested"), (1, "chunk0 M-tile0 MFMAs issued"), (2, "chunk0 M-tile1 MFMAs issued"), (3, "chunk0 sums in LDS"),
           (5, "chunk0 epilogue done"), (6, "end")) if st[i]})
f = [buf[i] for i in range(16)]
print("front: utterance WG0 chain (us):", round((f[7] - f[12]) / 100.0, 2), " first worker start/end rel. to WG0 entry:",
      round((f[8] - f[12]) / 100.0, 2), round((f[9] - f[12]) / 100.0, 2), " last WG start/end:", round((f[10] - f[12]) / 100.0, 2), round((f[11] - f[12]) / 100.0, 2))
print("gaps: front WG0 end -> lstm1 entry", (buf[16 + 4] - f[7]) / 100.0, " lstm1 end -> lstm2 entry", (buf[32 + 4] - buf[16 + 6]) / 100.0)

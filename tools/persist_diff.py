"""Diagnostic: where do the persistent decode launch and the launch path first differ (bitwise)?  python tools/persist_diff.py [B Tv steps]"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from test_gpu_parity import _full_case, _model
B, Tv, steps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 24, 6)
att = sys.argv[4] if len(sys.argv) > 4 else "SMA"
rate = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
hp, w, tokens, tl, mels, ml, masks, noise = _full_case(B, Tv, 40, steps, seed=8, att=att, rate=rate)
outs = {}
enc = gst = None
for flag in ("1", "0"):
    os.environ["GSTTACO_PERSIST_DECODE"] = flag
    if flag == "0" and os.environ.get("DIFF_FRONT"): os.environ["GSTTACO_FUSED_FRONT"] = os.environ["DIFF_FRONT"]
    gc.collect()
    m = _model(hp, w, B, Tv, 41)
    if enc is None:
        enc = m.encode(tokens).cpu().numpy().copy(); gst = m.Inference_GST_Step(mels, ml).cpu().numpy().copy()
    pre, stop, align = m.decode(enc, gst, masks, noise, steps=steps)
    torch.cuda.synchronize()
    print("persist", flag, "counters", m.decode_counters(), "err", m.handoff_error())
    outs[flag] = (pre.cpu().numpy(), stop.cpu().numpy(), align.cpu().numpy())
    del m
a, b = outs["1"], outs["0"]
r = a[0].shape[1] // steps
for t in range(steps):
    da = np.abs(a[2][:, t] - b[2][:, t]).max(); dp = np.abs(a[0][:, t * r:(t + 1) * r] - b[0][:, t * r:(t + 1) * r]).max(); ds = np.abs(a[1][:, t] - b[1][:, t]).max()
    print("step", t, "align diff %.3e  pre diff %.3e  stop diff %.3e" % (da, dp, ds))
    if da > 0 and t < 3:
        d = np.abs(a[2][:, t] - b[2][:, t]); i = np.unravel_index(d.argmax(), d.shape)
        print("   align: utterances differing", (d.max(1) > 0).sum(), "of", B, "worst at", i, a[2][:, t][i], b[2][:, t][i], "n positions", (d > 0).sum())
    if max(da, dp, ds) > 0 and t > 3:
        break

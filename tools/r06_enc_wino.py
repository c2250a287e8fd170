"""Round 6: the text encoder (conv stack + BiLSTM) at configs[1] with its five-tap layers on the split-bf16 Winograd kernel
(GSTTACO_ENC_WINO = 2 / 4) against the implicit GEMM (0): ms per encode, same box, alternating; and the encodings' difference."""
import os, sys, gc
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
hp = synthetic.config_hp("cfg2"); w = weights.synthetic_weights(hp, seed=0)
tokens, tl = synthetic.make_tokens(np.random.default_rng(1), 32, 128)
outs = {}
for rnd in range(2):
    for mode in ("0", "2", "4"):
        os.environ["GSTTACO_ENC_WINO"] = mode
        m = None; gc.collect()
        m = GST_Tacotron(hyper_parameters=hp, max_batch=32, max_tokens=128, max_ref_frames=4); m.Restore(weights=w)
        tok = torch.as_tensor(tokens, device="cuda")
        for _ in range(3): e = m.encode(tok)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m.encode(tok)
        e1.record(); torch.cuda.synchronize()
        outs[mode] = e.cpu().numpy()
        print("GSTTACO_ENC_WINO", mode, "encode ms %.4f" % (e0.elapsed_time(e1) / 20), "handoff err", m.handoff_error(), flush=True)
for mode in ("2", "4"):
    print("max |encoding(%s) - encoding(0)| = %.3g" % (mode, np.abs(outs[mode] - outs["0"]).max()))

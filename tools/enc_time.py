"""Encoder time (conv stack + BiLSTM), persistent vs per-step BiLSTM, at configs[1] (32 utterances x 128 tokens)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
hp = synthetic.config_hp("cfg2"); w = weights.synthetic_weights(hp, seed=0)
import gc
m = None
for mode in ("0", "1", "0", "1"):
    os.environ["GSTTACO_BILSTM_PERSIST"] = mode
    m = None
    gc.collect()            # the persistent BiLSTM is taken only by a process's only live context
    m = GST_Tacotron(hyper_parameters=hp, max_batch=32, max_tokens=128, max_ref_frames=4); m.Restore(weights=w)
    tokens, tl = synthetic.make_tokens(np.random.default_rng(1), 32, 128)
    tok = torch.as_tensor(tokens, device="cuda")
    for _ in range(3): m.encode(tok)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): m.encode(tok)
    e1.record(); torch.cuda.synchronize()
    print("persist", mode, "encode ms", e0.elapsed_time(e1) / 20, "err", m.handoff_error())

"""Postnet alone at configs[1] (32 utterances x 1000 frames; POSTNET_B / POSTNET_MIXED for the configs[4] shard's 64 x 1000 bf16): ms per call,
for A/B runs of the conv kernels (GSTTACO_LIB, GSTTACO_WINO, GSTTACO_WINO_SPLIT ...)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
B = int(os.environ.get("POSTNET_B", "32"))
hp = synthetic.config_hp("cfg2"); hp["Use_Mixed_Precision"] = os.environ.get("POSTNET_MIXED", "0") == "1"
w = weights.synthetic_weights(hp, seed=0)
m = GST_Tacotron(hyper_parameters=hp, max_batch=B, max_tokens=8, max_ref_frames=4); m.Restore(weights=w)
x = torch.as_tensor(np.clip(np.random.default_rng(0).normal(0, 1.5, (B, 1000, 80)), -4, 4).astype(np.float32), device="cuda")
for _ in range(3): y = m.postnet(x)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): m.postnet(x)
e1.record(); torch.cuda.synchronize()
print("postnet ms", round(e0.elapsed_time(e1) / 20, 4), "batch", B, "mixed", hp["Use_Mixed_Precision"], "checksum %.6f" % float(y.double().abs().mean()),
      {k: v for k, v in os.environ.items() if k.startswith("GSTTACO_")})

"""Round 6: the persistent BiLSTM's tagged-state hand-off under repetition and foreign load.  N encodes per shape on the persistent launch
against the per-step kernels' encodings of the same inputs (computed first, GSTTACO_BILSTM_PERSIST=0), bitwise, while a second stream keeps
the GPU busy with large GEMMs of another library: a stale or torn word in the state ring would show as a difference, a lost one as a give-up.
    python tools/bilstm_stress.py [calls per shape]"""
import gc, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
hp = synthetic.config_hp("cfg2"); w = weights.synthetic_weights(hp, seed=3)
rng = np.random.default_rng(7)
shapes = [(32, 128, False), (5, 37, True), (64, 60, True), (17, 128, False)]
inputs = []
for B, Tv, masked in shapes:
    tokens, _ = synthetic.make_tokens(rng, B, Tv)
    tl = rng.integers(1, Tv + 1, B).astype(np.int32) if masked else None
    inputs.append((torch.as_tensor(tokens, device="cuda"), None if tl is None else torch.as_tensor(tl, device="cuda")))
refs = []
os.environ["GSTTACO_BILSTM_PERSIST"] = "0"
m = GST_Tacotron(hyper_parameters=hp, max_batch=64, max_tokens=128, max_ref_frames=4); m.Restore(weights=w)
for tok, tl in inputs:
    refs.append(m.encode(tok, tl).clone())
torch.cuda.synchronize()
del m; gc.collect()
os.environ["GSTTACO_BILSTM_PERSIST"] = "1"
m = GST_Tacotron(hyper_parameters=hp, max_batch=64, max_tokens=128, max_ref_frames=4); m.Restore(weights=w)
stop = threading.Event(); side = torch.cuda.Stream()
def foreign():
    a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
    with torch.cuda.stream(side):
        while not stop.is_set():
            for _ in range(8): a @ b
            side.synchronize(); time.sleep(0.002)
th = threading.Thread(target=foreign); th.start()
bad = 0; t0 = time.perf_counter()
try:
    for i in range(N):
        for k, (tok, tl) in enumerate(inputs):
            e = m.encode(tok, tl)
            if not torch.equal(e, refs[k]): bad += 1
    torch.cuda.synchronize()
finally:
    stop.set(); th.join()
print("persistent BiLSTM (tagged state) vs per-step kernels: %d encodes over %d shapes under a foreign GEMM stream, %d differ, hand-off error %d, counters %s, %.1f s"
      % (N * len(inputs), len(inputs), bad, m.handoff_error(), m.debug_counters(), time.perf_counter() - t0))
sys.exit(1 if bad or m.handoff_error() else 0)

"""Experiment: does a postnet running on a second stream hide behind the latency-bound decode loop?
Stream A: whole Inference_Step; stream B (second context, same weights): postnet-only calls on 32 x F frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
hp, inputs = synthetic.config_inputs("cfg2", batch=32)
w = weights.synthetic_weights(hp, seed=0)
mk = lambda: GST_Tacotron(hyper_parameters=hp, max_batch=32, max_tokens=128, max_ref_frames=257).Restore(weights=w)
A, Bm = mk(), mk()
dev = A.device
tok = torch.as_tensor(inputs["tokens"]).to(dev); mels = torch.as_tensor(inputs["mels_for_gst"]).to(dev); lens = torch.as_tensor(inputs["mel_lengths_for_gst"]).to(dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def run_a(n):
    with torch.cuda.stream(sa):
        for i in range(n):
            A.Inference_Step(tok, None, None, mels, lens, seed=i)
for F in (1000, 500, 250):
    pre = torch.randn(32, F, 80, device=dev)
    def run_b(n):
        with torch.cuda.stream(sb):
            for i in range(n):
                Bm.postnet(pre)
    run_a(2); run_b(2); torch.cuda.synchronize()
    K = 8
    t0 = time.perf_counter(); run_a(K); torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / K
    NB = 8 * 1000 // F
    t0 = time.perf_counter(); run_b(NB); torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / NB
    # both: K inference steps with NB*K/8... postnets alongside: one full postnet's worth of frames per inference step
    per = 1000 // F
    t0 = time.perf_counter(); run_a(K); run_b(K * per); torch.cuda.synchronize(); tab = (time.perf_counter() - t0) / K
    print("F=%d: inference alone %.2f ms, postnet(F) alone %.3f ms (x%d = %.2f ms per 1000 frames), both interleaved %.2f ms per inference "
          "(sum would be %.2f)" % (F, 1e3 * ta, 1e3 * tb, per, 1e3 * tb * per, 1e3 * tab, 1e3 * (ta + tb * per)), flush=True)

"""Experiment: S independent Inference_Steps in flight on S HIP streams (S contexts) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
hp, inputs = synthetic.config_inputs("cfg2", batch=32)
w = weights.synthetic_weights(hp, seed=0)
for S in [int(x) for x in (sys.argv[1:] or ["1", "2", "3", "4"])]:
    models = [GST_Tacotron(hyper_parameters=hp, max_batch=32, max_tokens=128, max_ref_frames=257).Restore(weights=w) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    dev = models[0].device
    tok = torch.as_tensor(inputs["tokens"]).to(dev); mels = torch.as_tensor(inputs["mels_for_gst"]).to(dev); lens = torch.as_tensor(inputs["mel_lengths_for_gst"]).to(dev)
    def run(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % S]):
                models[i % S].Inference_Step(tok, None, None, mels, lens, seed=i)
        torch.cuda.synchronize()
    run(2 * S)
    K = 12
    t0 = time.perf_counter(); run(K); dt = time.perf_counter() - t0
    print("streams", S, "ms per Inference_Step %.2f" % (1e3 * dt / K), "frames/s %.0f" % (32 * 1000 * K / dt), flush=True)
    del models

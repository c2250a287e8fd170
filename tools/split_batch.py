"""Experiment: ONE batch of 32 utterances decoded as S independent sub-batches of 32/S on S streams (S contexts), against the
whole batch on one stream: does splitting the (independent) utterances into concurrent decode loops pay inside one call?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gst_tacotron_amd import synthetic, weights
from gst_tacotron_amd.model import GST_Tacotron
hp, inputs = synthetic.config_inputs("cfg2", batch=32)
w = weights.synthetic_weights(hp, seed=0)
for S in (1, 2, 4):
    b = 32 // S
    models = [GST_Tacotron(hyper_parameters=hp, max_batch=b, max_tokens=128, max_ref_frames=257).Restore(weights=w) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    dev = models[0].device
    tok = torch.as_tensor(inputs["tokens"]).to(dev); mels = torch.as_tensor(inputs["mels_for_gst"]).to(dev); lens = torch.as_tensor(inputs["mel_lengths_for_gst"]).to(dev)
    parts = [(tok[i * b:(i + 1) * b].contiguous(), mels[i * b:(i + 1) * b].contiguous(), lens[i * b:(i + 1) * b].contiguous()) for i in range(S)]
    def run(n):
        for it in range(n):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    models[i].Inference_Step(parts[i][0], None, None, parts[i][1], parts[i][2], seed=it)
        torch.cuda.synchronize()
    run(3)
    K = 8
    t0 = time.perf_counter(); run(K); dt = time.perf_counter() - t0
    print("batch 32 as %d x %d on %d streams: %.2f ms per 32 utterances -> %.0f frames/s" % (S, b, S, 1e3 * dt / K, 32 * 1000 * K / dt), flush=True)
    del models

"""Generates tests/golden/*.npz (run from the repo root: ``python -m oracle.gen_golden``).

PARITY UNPINNED: the vectors come from oracle/oracle_np.py in float64 (cross-checked
against oracle/torch_ref.py to <=1e-9 before they are written), NOT from TensorFlow --
the reference cannot be executed in the build container.

Fixtures hold inputs + expected outputs only; weights are regenerated from
(hyper-parameters, seed) by gst_tacotron_amd.weights.synthetic_weights and verified
through the per-tensor checksums stored next to the vectors.
"""
import json
import os
import sys

import numpy as np
import torch

from gst_tacotron_amd import synthetic, weights
from oracle import oracle_np, torch_ref

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def weight_checksums(w):
    names = sorted(w)
    return np.array([[float(np.sum(w[n], dtype=np.float64)), float(np.sum(np.abs(w[n]), dtype=np.float64))]
                     for n in names])


def make_case(name, hp, wseed, iseed, B, Tv, Tref, ref_lengths=None, token_lengths=None, steps=None):
    from gst_tacotron_amd.hparams import Dims
    d = Dims(hp)
    w = weights.synthetic_weights(hp, seed=wseed)
    rng = np.random.default_rng(iseed)
    tokens, tl = synthetic.make_tokens(rng, B, Tv, lengths=token_lengths)
    if d.gst:
        mels, ml = synthetic.make_ref_mels(rng, B, Tref, mel=d.mel, lengths=ref_lengths)
    else:
        mels, ml = None, None
    steps = steps or d.steps
    masks, noise = synthetic.make_randomness(rng, steps, B, Tv, d.prenet, rate=d.prenet_rate)
    o = oracle_np.inference_step(hp, w, tokens, mels, ml, masks, noise, steps=steps, dt=np.float64, with_vocoder=d.vocoder)
    t = torch_ref.TorchReference(hp, w, torch.float64).inference_step(tokens, mels, ml, masks, noise, steps=steps,
                                                                      with_vocoder=d.vocoder)
    checks = [(o[0], t[0], "mel"), (o[1], t[1], "stop"), (o[3], t[3], "align")]
    if d.vocoder:
        checks.append((o[2], t[2], "spectrogram"))
    for a, b, what in checks:
        err = np.abs(a - b.numpy()).max()
        assert err < 1e-9, (name, what, err)
    out = {
        "hp_json": np.array(json.dumps(hp)), "wseed": np.array(wseed), "steps": np.array(steps),
        "weight_checksums": weight_checksums(w),
        "tokens": tokens, "token_lengths": tl,
        "prenet_masks": masks.astype(np.uint8), "attn_noise": noise,
        "mels": o[0].astype(np.float32), "stops": o[1].astype(np.float32),
        "alignments": o[3].astype(np.float32), "pre_mel": o[4]["pre_mel"].astype(np.float32),
        "encoder": o[4]["encoder"].astype(np.float32),
    }
    if d.vocoder:
        out["spectrograms"] = o[2].astype(np.float32)
    if d.gst:
        out["mels_for_gst"] = mels
        out["mel_lengths_for_gst"] = ml
        out["gst"] = o[4]["gst"].astype(np.float32)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    # (i) tiny-dims full pipeline: every code path (r=1/2, GST on/off, SMA/BMA, ragged ref lengths)
    make_case("tiny_sma_r2_gst", synthetic.tiny_hp("SMA", r=2, gst=True), 3, 5, B=3, Tv=12, Tref=70,
              ref_lengths=np.array([70, 33, 64]))
    make_case("tiny_bma_r1_gst", synthetic.tiny_hp("BMA", r=1, gst=True, max_step=20), 4, 6, B=2, Tv=21, Tref=130,
              ref_lengths=np.array([129, 65]))
    make_case("tiny_sma_r1_nogst", synthetic.tiny_hp("SMA", r=1, gst=False, max_step=16), 5, 7, B=5, Tv=9, Tref=0)
    make_case("tiny_bma_r3_nodrop", synthetic.tiny_hp("BMA", r=3, gst=True, max_step=18, prenet_rate=0.0), 6, 8,
              B=2, Tv=17, Tref=64)
    # extension A13: step-wise location-sensitive attention (softmax and smoothing variants)
    lsa = synthetic.tiny_hp("SMA", r=2, gst=True, max_step=16)
    lsa["Tacotron2"]["Decoder"]["Attention"] = {"Type": "LSA", "Size": 16, "Conv": {"Filters": 8, "Kernel_Size": 7}}
    make_case("tiny_lsa_r2_gst", lsa, 9, 4, B=3, Tv=13, Tref=70)
    # (iii) full-dims short trajectory (LJSpeech hparams): B=2, T_v=16, 20 steps, r=2
    hp = synthetic.config_hp("cfg2")
    make_case("full_sma_r2_short", hp, 0, 11, B=2, Tv=16, Tref=96, ref_lengths=np.array([96, 50]), steps=20)
    hp1 = synthetic.config_hp("cfg1")
    make_case("full_cfg1_short", hp1, 0, 12, B=1, Tv=32, Tref=0, steps=12)


if __name__ == "__main__":
    sys.exit(main())

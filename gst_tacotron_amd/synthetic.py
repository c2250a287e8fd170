"""Seeded synthetic inputs for the BASELINE.json configs.

No datasets or checkpoints are reachable (no network), so bench and parity tests use
synthetic token / reference-mel batches of the shapes the reference's Feeder produces
(reference Feeder.py:161-227): tokens ``<S>`` ... ``<E>`` padded with ``<E>``=1,
``mels_for_gst`` with a prepended zero frame and values in [-4, 4].
"""
import copy

import numpy as np

from .hparams import load_hp

S_TOKEN, E_TOKEN = 0, 1


def make_tokens(rng, batch, n_tokens, vocab=34, lengths=None):
    """[B, n_tokens] int32: <S>, uniform ids in [2, vocab), <E>, padded with <E>."""
    tokens = np.full((batch, n_tokens), E_TOKEN, dtype=np.int32)
    if lengths is None:
        lengths = np.full((batch,), n_tokens, dtype=np.int32)
    for b in range(batch):
        n = int(lengths[b])
        tokens[b, 0] = S_TOKEN
        tokens[b, 1:n - 1] = rng.integers(2, vocab, n - 2)
        tokens[b, n - 1] = E_TOKEN
    return tokens, np.asarray(lengths, dtype=np.int32)


def make_ref_mels(rng, batch, t_ref, mel=80, lengths=None):
    """mels_for_gst [B, t_ref+1, mel]: frame 0 zeros, rest clip(N(0,1.5), -4, 4), zero padded."""
    if lengths is None:
        lengths = np.full((batch,), t_ref, dtype=np.int32)
    mels = np.zeros((batch, t_ref + 1, mel), dtype=np.float32)
    for b in range(batch):
        n = int(lengths[b])
        mels[b, 1:n + 1] = np.clip(rng.normal(0.0, 1.5, (n, mel)), -4.0, 4.0)
    return mels, np.asarray(lengths, dtype=np.int32)


def make_randomness(rng, steps, batch, t_v, prenet_sizes, rate=0.5):
    """Injected randomness for parity runs: prenet keep-masks
    [steps, n_prenet, B, size] (requires equal prenet sizes) and attention noise [steps, B, T_v]."""
    if len(set(prenet_sizes)) != 1:
        raise ValueError("injected prenet masks need equal prenet layer sizes")
    masks = (rng.random((steps, len(prenet_sizes), batch, prenet_sizes[0])) >= rate).astype(np.float32)
    noise = rng.standard_normal((steps, batch, t_v)).astype(np.float32)
    return masks, noise


def config_hp(name):
    """Hyper-parameter dicts for the BASELINE.json configs."""
    hp = load_hp()
    if name == "cfg1":        # TF2 CPU reference case: GST off, r=1, Max_Step 200
        hp["GST"]["Use"] = False
        hp["Step_Reduction"] = 1
        hp["Max_Step"] = 200
    elif name == "cfg2":      # headline: GST on, r=2, Max_Step 1000
        hp["Step_Reduction"] = 2
        hp["Max_Step"] = 1000
    else:
        raise KeyError(name)
    return hp


def config_inputs(name, batch=None, seed=None):
    """(hp, dict of inputs) for a BASELINE config; randomness for parity is made separately."""
    hp = config_hp(name)
    if name == "cfg1":
        rng = np.random.default_rng(0 if seed is None else seed)
        tokens, lens = make_tokens(rng, batch or 1, 32)
        return hp, {"tokens": tokens, "token_lengths": lens,
                    "initial_mels": np.zeros((tokens.shape[0], 1, 80), np.float32)}
    rng = np.random.default_rng(1 if seed is None else seed)
    B = batch or 32
    tokens, lens = make_tokens(rng, B, 128)
    mels, mel_lens = make_ref_mels(rng, B, 256)
    return hp, {"tokens": tokens, "token_lengths": lens,
                "initial_mels": np.zeros((B, 1, 80), np.float32),
                "mels_for_gst": mels, "mel_lengths_for_gst": mel_lens}


def tiny_hp(att_type="SMA", r=2, gst=True, max_step=24, prenet_rate=0.5):
    """Small-dimension config that still exercises every code path (all dims % 16 == 0)."""
    hp = copy.deepcopy(load_hp())
    hp["Sound"]["Mel_Dim"] = 16
    hp["GST"]["Use"] = gst
    ref = hp["GST"]["Reference_Encoder"]
    ref["Conv"]["Filters"] = [4, 4, 8, 8, 16, 16]
    ref["RNN"]["Size"] = 16
    ref["Dense"]["Size"] = 16
    st = hp["GST"]["Style_Token"]
    st["Size"] = 6
    st["Embedding"]["Size"] = 32
    st["Attention"] = {"Head": 4, "Size": 16}
    enc = hp["Tacotron2"]["Encoder"]
    enc["Embedding"]["Size"] = 32
    enc["Conv"]["Filters"] = [32, 32, 32]
    enc["RNN"]["Size"] = 16
    dec = hp["Tacotron2"]["Decoder"]
    dec["Prenet"] = {"Size": [32, 32], "Dropout_Rate": prenet_rate}
    dec["RNN"]["Size"] = [64, 64]
    dec["Attention"] = {"Type": att_type, "Size": 16}
    dec["Conv"]["Filters"] = [32, 32, 32, 32]
    hp["Sound"]["Spectrogram_Dim"] = 33            # n_fft 64; like the real 513 not a multiple of 4
    hp["Sound"]["Frame_Length"], hp["Sound"]["Frame_Shift"] = 64, 16
    hp["Vocoder_Taco1"]["CBHG"] = {"Conv_Bank": {"Stack_Count": 4, "Filters": 16}, "Pool": {"Pool_Size": 2, "Strides": 1},
                                   "Conv1D": {"Filters": [32, 32], "Kernel_Size": [3, 3]},
                                   "Highwaynet": {"Count": 2, "Size": 32}, "RNN": {"Size": 16, "Zoneout": 0.0}}
    hp["Step_Reduction"] = r
    hp["Max_Step"] = max_step
    return hp

"""Hyper_Parameters.json config surface (same schema as the reference).

The reference reads ``Hyper_Parameters.json`` from the CWD at import time in five
modules (reference Model.py:17, Modules/Taco2.py:6-10, Modules/GST.py:6-10,
Feeder.py:10).  Here the same JSON schema is loaded explicitly (path or dict) and
flattened into the handful of integers the C-ABI ``gsttaco_config`` needs.

Additive keys (absent in the reference, defaults reproduce it):
  Tacotron2.Decoder.Attention.Sigmoid_Noise  -- SMA default 2.0 (Steps.py:212),
                                                 BMA default 0.0 (Steps.py:58)
"""
import copy
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_HP_PATH = os.path.join(_HERE, "Hyper_Parameters.json")
DEFAULT_TOKEN_PATH = os.path.join(_HERE, "Token_Index_Dict.ENG.json")

# reference Taco2.py:66-75 accepts exactly "BMA"/"SMA".  "LSA" is an EXTENSION: north_star names a
# location-sensitive attention, which the reference only has as a whole-sequence Keras layer
# (Modules/Attention/Layers.py:289-444) that its decoder cannot select; it is restated step-wise here.
ATTENTION_TYPES = ("BMA", "SMA", "LSA")


def load_hp(source=None):
    """Return the hyper-parameter dict.  ``source``: None (packaged LJSpeech
    defaults), a path to a Hyper_Parameters.json, or an already-loaded dict."""
    if source is None:
        source = DEFAULT_HP_PATH
    if isinstance(source, dict):
        return copy.deepcopy(source)
    with open(source, "r") as f:
        return json.load(f)


def load_token_dict(hp=None, base_dir=None):
    path = DEFAULT_TOKEN_PATH
    if hp is not None and hp.get("Token_JSON_Path"):
        cand = hp["Token_JSON_Path"]
        for p in (cand, os.path.join(base_dir or ".", cand), os.path.join(_HERE, cand)):
            if os.path.exists(p):
                path = p
                break
    with open(path, "r") as f:
        return json.load(f)


def attention_sigmoid_noise(hp):
    att = hp["Tacotron2"]["Decoder"]["Attention"]
    if att["Type"] == "LSA":
        return 0.0
    if "Sigmoid_Noise" in att:
        return float(att["Sigmoid_Noise"])
    return 2.0 if att["Type"] == "SMA" else 0.0


class Dims:
    """Flat view of the hot-path dimensions."""

    def __init__(self, hp, vocab=None):
        t2 = hp["Tacotron2"]
        enc, dec = t2["Encoder"], t2["Decoder"]
        self.mel = int(hp["Sound"]["Mel_Dim"])
        self.r = int(hp["Step_Reduction"])
        self.max_step = int(hp["Max_Step"])
        self.steps = self.max_step // self.r          # Taco2.py:213
        self.vocab = int(vocab) if vocab is not None else len(load_token_dict(hp))
        self.emb = int(enc["Embedding"]["Size"])
        self.enc_filters = [int(x) for x in enc["Conv"]["Filters"]]
        self.enc_kernels = [int(x) for x in enc["Conv"]["Kernel_Size"]]
        if any(int(s) != 1 for s in enc["Conv"]["Strides"]):
            raise ValueError("Encoder conv strides other than 1 are not supported")
        self.enc_rnn = int(enc["RNN"]["Size"])
        self.enc_out = 2 * self.enc_rnn
        self.prenet = [int(x) for x in dec["Prenet"]["Size"]]
        self.prenet_rate = float(dec["Prenet"]["Dropout_Rate"])
        self.dec_rnn = [int(x) for x in dec["RNN"]["Size"]]
        self.att_type = dec["Attention"]["Type"]
        if self.att_type not in ATTENTION_TYPES:
            # same error text as reference Taco2.py:75
            raise ValueError("Unsupported attention type: {}".format(self.att_type))
        self.att = int(dec["Attention"]["Size"])
        self.sigmoid_noise = attention_sigmoid_noise(hp)
        # LSA extension hyper-parameters (new, additive keys: Attention.Conv.{Filters,Kernel_Size}, Cumulate_Weights, Smoothing)
        conv = dec["Attention"].get("Conv", {})
        self.loc_filters = int(conv.get("Filters", 32))
        self.loc_kernel = int(conv.get("Kernel_Size", 31))
        self.lsa_cumulate = bool(dec["Attention"].get("Cumulate_Weights", True))   # Layers.py:302
        self.lsa_smoothing = bool(dec["Attention"].get("Smoothing", False))        # Layers.py:300
        self.post_filters = [int(x) for x in dec["Conv"]["Filters"]] + [self.mel]  # Taco2.py:133
        self.post_kernels = [int(x) for x in dec["Conv"]["Kernel_Size"]] + [5]     # Taco2.py:134
        self.post_tanh = len(dec["Conv"]["Filters"]) - 1                           # Taco2.py:145 (F9)
        self.gst = bool(hp["GST"]["Use"])
        if self.gst:
            g = hp["GST"]
            ref = g["Reference_Encoder"]
            self.ref_filters = [int(x) for x in ref["Conv"]["Filters"]]
            self.ref_kernels = [int(x) for x in ref["Conv"]["Kernel_Size"]]
            self.ref_strides = [int(x) for x in ref["Conv"]["Strides"]]
            self.ref_rnn = int(ref["RNN"]["Size"])
            self.ref_dense = int(ref["Dense"]["Size"])
            st = g["Style_Token"]
            self.n_tokens = int(st["Size"])
            self.token_emb = int(st["Embedding"]["Size"])
            self.heads = int(st["Attention"]["Head"])
            self.gst_att = int(st["Attention"]["Size"])
            if self.gst_att % self.heads != 0:
                # same check as reference Layers.py:155-156
                raise ValueError("size must be divisible by num_heads. ('{}' % '{}' != 0)".format(
                    self.gst_att, self.heads))
            self.ref_freq = self.mel
            for k, s in zip(self.ref_kernels, self.ref_strides):
                self.ref_freq = -(-self.ref_freq // s)
            self.ref_stride_prod = 1
            for s in self.ref_strides:
                self.ref_stride_prod *= s
            self.gru_in = self.ref_freq * self.ref_filters[-1]
            self.mem = self.gst_att + self.enc_out      # GST.py:121-124: [gst | enc]
        else:
            self.mem = self.enc_out
        self.proj_out = self.mel * self.r + 1           # Taco2.py:88
        # Sound: audio front / back end (reference Audio.py, Pattern_Generator.py:39-60): SURVEY rows N2 / N4
        snd = hp["Sound"]
        self.spec = int(snd.get("Spectrogram_Dim", 0))
        self.sample_rate = int(snd.get("Sample_Rate", 0))
        self.frame_length = int(snd.get("Frame_Length", 0))
        self.frame_shift = int(snd.get("Frame_Shift", 0))
        self.max_abs_mel = float(snd.get("Max_Abs_Mel") or 0.0)
        self.audio = self.spec > 1 and self.sample_rate > 0 and self.frame_length > 0 and self.frame_shift > 0
        # CBHG vocoder (reference Taco2.py:234-260, 285-424): SURVEY row N1, optional third output of Inference_Step
        self.vocoder = "Vocoder_Taco1" in hp
        if self.vocoder:
            cb = hp["Vocoder_Taco1"]["CBHG"]
            self.spec = int(hp["Sound"]["Spectrogram_Dim"])
            self.bank_count = int(cb["Conv_Bank"]["Stack_Count"])
            self.bank_filters = int(cb["Conv_Bank"]["Filters"])
            if int(cb["Pool"]["Pool_Size"]) != 2 or int(cb["Pool"]["Strides"]) != 1:
                raise ValueError("Vocoder_Taco1.CBHG.Pool other than Pool_Size 2 / Strides 1 is not supported")
            self.voc_proj_filters = [int(x) for x in cb["Conv1D"]["Filters"]]
            self.voc_proj_kernels = [int(x) for x in cb["Conv1D"]["Kernel_Size"]]
            self.highway_count = int(cb["Highwaynet"]["Count"])
            self.highway_size = int(cb["Highwaynet"]["Size"])
            self.voc_rnn = int(cb["RNN"]["Size"])

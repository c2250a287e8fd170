"""ctypes binding of include/gsttaco.h (the C-ABI of the HIP hot path).

This is the stub a maintainer of the reference would add in place of the Keras
functional model (reference Model.py:145-156); see INTEGRATION.md.  There is no CPU
fallback: a missing library raises ImportError here and a missing GPU surfaces as
GSTTACO_E_NO_DEVICE from gsttaco_finalize_weights.
"""
import ctypes
import os

import numpy as np
import torch  # noqa: F401  -- must be imported BEFORE the dlopen below: torch bundles the HIP runtime that owns the
#                     tensors/streams handed to the C-ABI, and libgsttaco.so has to bind to that same libamdhip64

from .hparams import Dims

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgsttaco.so")

MAX_LAYERS = 8
ABI_VERSION = 12
ATT_CODES = {"BMA": 0, "SMA": 1, "LSA": 2}

# every symbol include/gsttaco.h declares
EXPORTED_SYMBOLS = (
    "gsttaco_abi_version", "gsttaco_create", "gsttaco_destroy", "gsttaco_last_error",
    "gsttaco_num_weights", "gsttaco_weight_info", "gsttaco_load_weight", "gsttaco_finalize_weights",
    "gsttaco_encode", "gsttaco_gst", "gsttaco_decode", "gsttaco_postnet", "gsttaco_vocoder", "gsttaco_inference_step",
    "gsttaco_mel_frontend", "gsttaco_mel_basis", "gsttaco_griffin_lim", "gsttaco_crc32c",
    "gsttaco_set_profiling", "gsttaco_get_profile", "gsttaco_lstm_launch_bytes", "gsttaco_debug_stamps", "gsttaco_decode_plan", "gsttaco_debug_randomness",
    "gsttaco_set_graph_policy", "gsttaco_graph_cache_size", "gsttaco_debug_handoff_error", "gsttaco_debug_raise_handoff_error", "gsttaco_debug_counters",
    "gsttaco_synchronize",
)

_I32A = ctypes.c_int32 * MAX_LAYERS


class Config(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("device", ctypes.c_int32),
        ("mel_dim", ctypes.c_int32), ("step_reduction", ctypes.c_int32), ("max_step", ctypes.c_int32),
        ("vocab", ctypes.c_int32), ("emb", ctypes.c_int32), ("n_enc_conv", ctypes.c_int32),
        ("enc_filters", _I32A), ("enc_kernels", _I32A), ("enc_rnn", ctypes.c_int32),
        ("n_prenet", ctypes.c_int32), ("prenet", _I32A), ("prenet_rate", ctypes.c_float),
        ("n_dec_rnn", ctypes.c_int32), ("dec_rnn", _I32A),
        ("att_type", ctypes.c_int32), ("att_size", ctypes.c_int32), ("sigmoid_noise", ctypes.c_float),
        ("loc_filters", ctypes.c_int32), ("loc_kernel", ctypes.c_int32), ("lsa_cumulate", ctypes.c_int32),
        ("lsa_smoothing", ctypes.c_int32),
        ("n_post", ctypes.c_int32), ("post_filters", _I32A), ("post_kernels", _I32A), ("post_tanh", ctypes.c_int32),
        ("gst_use", ctypes.c_int32), ("n_ref_conv", ctypes.c_int32),
        ("ref_filters", _I32A), ("ref_kernels", _I32A), ("ref_strides", _I32A),
        ("ref_rnn", ctypes.c_int32), ("ref_dense", ctypes.c_int32), ("n_tokens", ctypes.c_int32),
        ("token_emb", ctypes.c_int32), ("heads", ctypes.c_int32), ("gst_att", ctypes.c_int32),
        ("voc_use", ctypes.c_int32), ("spec_dim", ctypes.c_int32), ("bank_count", ctypes.c_int32),
        ("bank_filters", ctypes.c_int32), ("n_voc_proj", ctypes.c_int32), ("voc_proj_filters", _I32A),
        ("voc_proj_kernels", _I32A), ("highway_count", ctypes.c_int32), ("highway_size", ctypes.c_int32),
        ("voc_rnn", ctypes.c_int32),
        ("sample_rate", ctypes.c_int32), ("frame_length", ctypes.c_int32), ("frame_shift", ctypes.c_int32),
        ("max_abs_mel", ctypes.c_float), ("max_wav_samples", ctypes.c_int32),
        ("mixed_precision", ctypes.c_int32),
        ("max_batch", ctypes.c_int32), ("max_tokens", ctypes.c_int32), ("max_ref_frames", ctypes.c_int32),
    ]


class GstTacoError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("gsttaco error {}: {}".format(code, message))
        self.code = code


_lib = None


def load_library(path=None):
    """dlopen the C-ABI library; fails loudly if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # GSTTACO_LIB: an alternative build of the same ABI (A/B timing of two builds on one GPU box; tools/ab.sh)
    p = path or os.environ.get("GSTTACO_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            "{} not found: build the HIP extension first (python -m gst_tacotron_amd.build). "
            "There is no CPU fallback for the hot path.".format(p))
    lib = ctypes.CDLL(p)
    vp, i32, u64, f32p = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_float)
    lib.gsttaco_abi_version.restype = ctypes.c_int
    lib.gsttaco_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    lib.gsttaco_destroy.argtypes = [vp]
    lib.gsttaco_destroy.restype = None
    lib.gsttaco_last_error.argtypes = [vp]
    lib.gsttaco_last_error.restype = ctypes.c_char_p
    lib.gsttaco_num_weights.argtypes = [vp]
    lib.gsttaco_weight_info.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_char_p),
                                        ctypes.POINTER(ctypes.c_int64 * 4), ctypes.POINTER(ctypes.c_int)]
    lib.gsttaco_load_weight.argtypes = [vp, ctypes.c_char_p, f32p, ctypes.POINTER(ctypes.c_int64), i32]
    lib.gsttaco_finalize_weights.argtypes = [vp]
    lib.gsttaco_encode.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.gsttaco_gst.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.gsttaco_decode.argtypes = [vp, vp, vp, vp, vp, vp, u64, i32, i32, i32, vp, vp, vp, vp]
    lib.gsttaco_postnet.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.gsttaco_vocoder.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.gsttaco_inference_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.gsttaco_mel_frontend.argtypes = [vp, vp, vp, i32, i32, ctypes.c_float, vp, vp, i32, vp]
    lib.gsttaco_mel_basis.argtypes = [vp, f32p]
    lib.gsttaco_griffin_lim.argtypes = [vp, vp, vp, i32, i32, i32, ctypes.c_float, ctypes.c_float, vp, u64, vp, vp, i32, vp]
    lib.gsttaco_set_profiling.argtypes = [vp, i32]
    lib.gsttaco_get_profile.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
    lib.gsttaco_lstm_launch_bytes.argtypes = [vp, i32, i32]
    lib.gsttaco_lstm_launch_bytes.restype = ctypes.c_int64
    lib.gsttaco_debug_stamps.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.gsttaco_debug_stamps.restype = ctypes.c_int
    lib.gsttaco_decode_plan.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_int32)]
    lib.gsttaco_decode_plan.restype = ctypes.c_int
    lib.gsttaco_debug_randomness.argtypes = [vp, vp, vp, i32, i32, i32]
    lib.gsttaco_debug_randomness.restype = ctypes.c_int
    lib.gsttaco_set_graph_policy.argtypes = [vp, i32, i32]
    lib.gsttaco_set_graph_policy.restype = ctypes.c_int
    lib.gsttaco_graph_cache_size.argtypes = [vp]
    lib.gsttaco_graph_cache_size.restype = ctypes.c_int
    lib.gsttaco_debug_handoff_error.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    lib.gsttaco_debug_handoff_error.restype = ctypes.c_int
    lib.gsttaco_debug_raise_handoff_error.argtypes = [vp, ctypes.c_uint32]
    lib.gsttaco_debug_raise_handoff_error.restype = ctypes.c_int
    lib.gsttaco_debug_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.gsttaco_debug_counters.restype = ctypes.c_int
    lib.gsttaco_synchronize.argtypes = [vp, vp]
    lib.gsttaco_synchronize.restype = ctypes.c_int
    for fn in ("gsttaco_create", "gsttaco_num_weights", "gsttaco_weight_info", "gsttaco_load_weight",
               "gsttaco_finalize_weights", "gsttaco_encode", "gsttaco_gst", "gsttaco_decode", "gsttaco_postnet", "gsttaco_vocoder",
               "gsttaco_inference_step", "gsttaco_set_profiling", "gsttaco_get_profile", "gsttaco_mel_frontend",
               "gsttaco_mel_basis", "gsttaco_griffin_lim"):
        getattr(lib, fn).restype = ctypes.c_int
    if path is None:
        _lib = lib
    return lib


def make_config(hp, vocab=None, device=0, max_batch=32, max_tokens=256, max_ref_frames=1025, max_wav_seconds=0.0):
    d = Dims(hp, vocab)
    c = Config()
    c.abi_version, c.device = ABI_VERSION, device
    c.mel_dim, c.step_reduction, c.max_step = d.mel, d.r, d.max_step
    c.vocab, c.emb, c.n_enc_conv = d.vocab, d.emb, len(d.enc_filters)
    if max(len(d.enc_filters), len(d.post_filters), len(d.prenet), len(d.dec_rnn)) > MAX_LAYERS:
        raise ValueError("too many layers for the C-ABI config (max {})".format(MAX_LAYERS))
    for i, (f, k) in enumerate(zip(d.enc_filters, d.enc_kernels)):
        c.enc_filters[i], c.enc_kernels[i] = f, k
    c.enc_rnn = d.enc_rnn
    c.n_prenet = len(d.prenet)
    for i, s in enumerate(d.prenet):
        c.prenet[i] = s
    c.prenet_rate = d.prenet_rate
    c.n_dec_rnn = len(d.dec_rnn)
    for i, s in enumerate(d.dec_rnn):
        c.dec_rnn[i] = s
    c.att_type, c.att_size, c.sigmoid_noise = ATT_CODES[d.att_type], d.att, d.sigmoid_noise
    c.loc_filters, c.loc_kernel = d.loc_filters, d.loc_kernel
    c.lsa_cumulate, c.lsa_smoothing = int(d.lsa_cumulate), int(d.lsa_smoothing)
    c.n_post = len(d.post_filters)
    for i, (f, k) in enumerate(zip(d.post_filters, d.post_kernels)):
        c.post_filters[i], c.post_kernels[i] = f, k
    c.post_tanh = d.post_tanh
    c.gst_use = int(d.gst)
    if d.gst:
        if len(d.ref_filters) > MAX_LAYERS:
            raise ValueError("too many reference-encoder layers")
        c.n_ref_conv = len(d.ref_filters)
        for i, (f, k, s) in enumerate(zip(d.ref_filters, d.ref_kernels, d.ref_strides)):
            c.ref_filters[i], c.ref_kernels[i], c.ref_strides[i] = f, k, s
        c.ref_rnn, c.ref_dense, c.n_tokens = d.ref_rnn, d.ref_dense, d.n_tokens
        c.token_emb, c.heads, c.gst_att = d.token_emb, d.heads, d.gst_att
    c.spec_dim = d.spec
    if d.audio:
        c.sample_rate, c.frame_length, c.frame_shift, c.max_abs_mel = d.sample_rate, d.frame_length, d.frame_shift, d.max_abs_mel
        c.max_wav_samples = int(max_wav_seconds * d.sample_rate)
    elif max_wav_seconds:
        raise ValueError("Hyper_Parameters has no complete Sound section for the audio entry points")
    c.voc_use = int(d.vocoder)
    if d.vocoder:
        if len(d.voc_proj_filters) > MAX_LAYERS:
            raise ValueError("too many vocoder projection layers")
        c.bank_count, c.bank_filters = d.bank_count, d.bank_filters
        c.n_voc_proj = len(d.voc_proj_filters)
        for i, (f, k) in enumerate(zip(d.voc_proj_filters, d.voc_proj_kernels)):
            c.voc_proj_filters[i], c.voc_proj_kernels[i] = f, k
        c.highway_count, c.highway_size, c.voc_rnn = d.highway_count, d.highway_size, d.voc_rnn
    c.mixed_precision = int(bool(hp.get("Use_Mixed_Precision", False)))
    c.max_batch, c.max_tokens, c.max_ref_frames = max_batch, max_tokens, max_ref_frames
    return c


class Context:
    """RAII wrapper around gsttaco_ctx."""

    def __init__(self, hp, vocab=None, device=0, max_batch=32, max_tokens=256, max_ref_frames=1025, lib=None,
                 max_wav_seconds=0.0):
        self.lib = lib or load_library()
        self.cfg = make_config(hp, vocab, device, max_batch, max_tokens, max_ref_frames, max_wav_seconds)
        self.handle = ctypes.c_void_p()
        rc = self.lib.gsttaco_create(ctypes.byref(self.cfg), ctypes.byref(self.handle))
        if rc != 0:
            raise GstTacoError(rc, self.lib.gsttaco_last_error(None).decode())

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.gsttaco_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise GstTacoError(rc, self.lib.gsttaco_last_error(self.handle).decode())

    def manifest(self):
        out = {}
        n = self.lib.gsttaco_num_weights(self.handle)
        for i in range(n):
            name, shape, nd = ctypes.c_char_p(), (ctypes.c_int64 * 4)(), ctypes.c_int()
            self.check(self.lib.gsttaco_weight_info(self.handle, i, ctypes.byref(name), ctypes.byref(shape), ctypes.byref(nd)))
            out[name.value.decode()] = tuple(shape[k] for k in range(nd.value))
        return out

    def load_weights(self, weights):
        for name in self.manifest():
            if name not in weights:
                raise KeyError("missing weight '{}'".format(name))
            a = np.asarray(weights[name], dtype=np.float32, order="C")
            if not a.flags["C_CONTIGUOUS"]:
                a = a.copy()
            shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            self.check(self.lib.gsttaco_load_weight(
                self.handle, name.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), shape, a.ndim))

    def finalize(self):
        self.check(self.lib.gsttaco_finalize_weights(self.handle))

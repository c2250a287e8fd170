"""Weight manifest for the inference hot path and a seeded
synthetic initialiser.  Shapes are the TF/Keras variable layouts the reference
creates (kernel ``[in, out]``, conv ``[k, Cin, Cout]`` / ``[kh, kw, Cin, Cout]``,
LSTM gate order i,f,c,o, GRU bias ``[2, 3u]``), so a converted reference
checkpoint drops in by name.

Files: a flat ``.npz`` (name -> float32 array).  The reference's
``tf.train.Checkpoint`` reader (Model.py:186-189, 267-276) is a later row (N3).
"""
import numpy as np

from .hparams import Dims

BN_FIELDS = ("gamma", "beta", "moving_mean", "moving_variance")


def manifest(hp, vocab=None):
    """Ordered dict name -> shape for every tensor on the hot path."""
    d = Dims(hp, vocab)
    m = {}
    # --- Tacotron-2 text encoder (reference Taco2.py:12-51)
    m["encoder.embedding"] = (d.vocab, d.emb)
    cin = d.emb
    for i, (f, k) in enumerate(zip(d.enc_filters, d.enc_kernels)):
        m[f"encoder.conv{i}.kernel"] = (k, cin, f)
        for b in BN_FIELDS:
            m[f"encoder.conv{i}.bn.{b}"] = (f,)
        cin = f
    for direction in ("fwd", "bwd"):
        m[f"encoder.bilstm.{direction}.kernel"] = (cin, 4 * d.enc_rnn)
        m[f"encoder.bilstm.{direction}.recurrent_kernel"] = (d.enc_rnn, 4 * d.enc_rnn)
        m[f"encoder.bilstm.{direction}.bias"] = (4 * d.enc_rnn,)
    # --- GST (reference GST.py:12-109, Layers.py:147-285)
    if d.gst:
        cin = 1
        for i, (f, k) in enumerate(zip(d.ref_filters, d.ref_kernels)):
            m[f"gst.ref.conv{i}.kernel"] = (k, k, cin, f)
            for b in BN_FIELDS:
                m[f"gst.ref.conv{i}.bn.{b}"] = (f,)
            cin = f
        m["gst.ref.gru.kernel"] = (d.gru_in, 3 * d.ref_rnn)
        m["gst.ref.gru.recurrent_kernel"] = (d.ref_rnn, 3 * d.ref_rnn)
        m["gst.ref.gru.bias"] = (2, 3 * d.ref_rnn)
        m["gst.ref.dense.kernel"] = (d.ref_rnn, d.ref_dense)
        m["gst.ref.dense.bias"] = (d.ref_dense,)
        m["gst.tokens"] = (d.n_tokens, d.token_emb)
        m["gst.mha.query.kernel"] = (d.ref_dense, d.gst_att)
        m["gst.mha.query.bias"] = (d.gst_att,)
        m["gst.mha.value.kernel"] = (d.token_emb, d.gst_att)
        m["gst.mha.value.bias"] = (d.gst_att,)
        m["gst.mha.ln.gamma"] = (d.gst_att,)
        m["gst.mha.ln.beta"] = (d.gst_att,)
    # --- decoder step (reference Taco2.py:53-120, Steps.py:65-105)
    cin = d.mel
    for i, s in enumerate(d.prenet):
        m[f"decoder.prenet{i}.kernel"] = (cin, s)
        m[f"decoder.prenet{i}.bias"] = (s,)
        cin = s
    m["decoder.attention.query.kernel"] = (d.prenet[-1], d.att)
    m["decoder.attention.query.bias"] = (d.att,)
    m["decoder.attention.value.kernel"] = (d.mem, d.att)
    m["decoder.attention.value.bias"] = (d.att,)
    if d.att_type == "LSA":     # extension A13: Layers.py:310-321, 335-341
        m["decoder.attention.location_conv.kernel"] = (d.loc_kernel, 1, d.loc_filters)
        m["decoder.attention.location_conv.bias"] = (d.loc_filters,)
        m["decoder.attention.location_dense.kernel"] = (d.loc_filters, d.att)
        m["decoder.attention.location_dense.bias"] = (d.att,)
        m["decoder.attention.bias"] = (d.att,)
    else:
        m["decoder.attention.v"] = (d.att,)
        m["decoder.attention.score_bias"] = ()
    cin = d.prenet[-1] + d.att
    for i, s in enumerate(d.dec_rnn):
        m[f"decoder.lstm{i}.kernel"] = (cin, 4 * s)
        m[f"decoder.lstm{i}.recurrent_kernel"] = (s, 4 * s)
        m[f"decoder.lstm{i}.bias"] = (4 * s,)
        cin = s
    m["decoder.projection.kernel"] = (d.dec_rnn[-1] + d.att, d.proj_out)
    m["decoder.projection.bias"] = (d.proj_out,)
    # --- postnet (reference Taco2.py:131-149)
    cin = d.mel
    for i, (f, k) in enumerate(zip(d.post_filters, d.post_kernels)):
        m[f"postnet.conv{i}.kernel"] = (k, cin, f)
        for b in BN_FIELDS:
            m[f"postnet.conv{i}.bn.{b}"] = (f,)
        cin = f
    # --- CBHG vocoder (reference Taco2.py:234-260, 285-424), SURVEY row N1
    if d.vocoder:
        for i in range(d.bank_count):
            m[f"vocoder.convbank{i}.kernel"] = (i + 1, d.mel, d.bank_filters)
            for b in BN_FIELDS:
                m[f"vocoder.convbank{i}.bn.{b}"] = (d.bank_filters,)
        cin = d.bank_count * d.bank_filters
        for i, (f, k) in enumerate(zip(d.voc_proj_filters, d.voc_proj_kernels)):
            m[f"vocoder.proj{i}.kernel"] = (k, cin, f)
            for b in BN_FIELDS:
                m[f"vocoder.proj{i}.bn.{b}"] = (f,)
            cin = f
        if cin != d.mel:                                    # Taco2.py:342-345
            m["vocoder.proj_dense.kernel"] = (cin, d.mel)
            m["vocoder.proj_dense.bias"] = (d.mel,)
        if d.mel != d.highway_size:                         # Taco2.py:348-351
            m["vocoder.highway_in.kernel"] = (d.mel, d.highway_size)
            m["vocoder.highway_in.bias"] = (d.highway_size,)
        for i in range(d.highway_count):
            for g in ("relu", "sigmoid"):
                m[f"vocoder.highway{i}.{g}.kernel"] = (d.highway_size, d.highway_size)
                m[f"vocoder.highway{i}.{g}.bias"] = (d.highway_size,)
        for direction in ("fwd", "bwd"):
            m[f"vocoder.bilstm.{direction}.kernel"] = (d.highway_size, 4 * d.voc_rnn)
            m[f"vocoder.bilstm.{direction}.recurrent_kernel"] = (d.voc_rnn, 4 * d.voc_rnn)
            m[f"vocoder.bilstm.{direction}.bias"] = (4 * d.voc_rnn,)
        m["vocoder.dense.kernel"] = (2 * d.voc_rnn, d.spec)
        m["vocoder.dense.bias"] = (d.spec,)
    return m


def synthetic_weights(hp, seed=0, vocab=None, gain=1.0):
    """Seeded synthetic weights: Glorot-uniform kernels, BN
    ``moving_variance`` in [0.5, 1.5], ``moving_mean`` ~ N(0, 0.1), unit forget
    bias, so activations stay O(1) over 1000 decode steps."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in manifest(hp, vocab).items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "gamma":
            a = rng.uniform(0.8, 1.2, shape)
        elif leaf == "beta":
            a = rng.normal(0.0, 0.1, shape)
        elif leaf == "moving_mean":
            a = rng.normal(0.0, 0.1, shape)
        elif leaf == "moving_variance":
            a = rng.uniform(0.5, 1.5, shape)
        elif leaf == "bias":
            a = rng.normal(0.0, 0.05, shape)
            if ".lstm" in name or "bilstm" in name:
                u = shape[0] // 4
                a[u:2 * u] += 1.0          # Keras unit_forget_bias
        elif leaf == "score_bias":
            a = np.asarray(rng.normal(0.0, 0.5))
        elif leaf == "tokens":
            a = np.clip(rng.normal(0.0, 0.5, shape), -1.0, 1.0)   # GST.py:87 TruncatedNormal(0.5)
        elif leaf == "embedding":
            a = rng.uniform(-0.5, 0.5, shape)
        elif leaf == "v":
            lim = np.sqrt(6.0 / (shape[0] + 1)) * 4.0
            a = rng.uniform(-lim, lim, shape)
        else:  # kernels: Glorot uniform over (fan_in, fan_out) incl. receptive field
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = rf * shape[-2], rf * shape[-1]
            lim = gain * np.sqrt(6.0 / (fan_in + fan_out))
            a = rng.uniform(-lim, lim, shape)
        w[name] = np.require(np.asarray(a, dtype=np.float32), requirements="C").reshape(shape)
    return w


def check_weights(hp, weights, vocab=None):
    """Raise KeyError/ValueError if ``weights`` does not match the manifest."""
    for name, shape in manifest(hp, vocab).items():
        if name not in weights:
            raise KeyError("missing weight '{}'".format(name))
        got = tuple(np.shape(weights[name]))
        if got != tuple(shape):
            raise ValueError("weight '{}' has shape {}, expected {}".format(name, got, tuple(shape)))


def save_npz(path, weights):
    np.savez(path, **{k: np.asarray(v, dtype=np.float32) for k, v in weights.items()})


def load_npz(path):
    with np.load(path) as z:
        return {k: np.asarray(z[k], dtype=np.float32, order="C") for k in z.files}

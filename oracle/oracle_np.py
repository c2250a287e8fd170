"""CPU ORACLE (test infrastructure, NOT product code) -- NumPy restatement of the
CODEJIN/GST_Tacotron inference hot path.

PARITY UNPINNED: the reference is TensorFlow-2/Keras code; TensorFlow is neither
installed nor installable in the build container and the reference ships no tests,
golden vectors or fixtures for this path.  This file restates
the algorithm from the reference source plus the documented Keras defaults listed in
SURVEY.md Appendix A; it is cross-checked against an independent torch-CPU
restatement (oracle/torch_ref.py) but has never been compared with real TF output.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The arithmetic lives in the un-vendored dependency ``tensorflow>=2.1.2``
(reference Requirements.txt:3, no pinned version).

Every function takes ``dt`` (np.float64 for the high-precision oracle, np.float32 to
see the fp32 noise floor) and cites the reference lines it follows.
"""
import numpy as np

BN_EPS = 1e-3   # tf.keras.layers.BatchNormalization default epsilon


# ----------------------------------------------------------------------------- helpers
# Mixed precision (Use_Mixed_Precision; BASELINE configs[4]): the HIP path rounds the operands of its MFMA GEMMs (Conv1D,
# LSTM gates, Value / projection / vocoder Dense layers) to bfloat16 and accumulates in fp32; everything else -- prenet,
# attention query and scores, GST, BN, activations, state -- stays fp32.  `mm` emulates exactly that when MIXED is set
# (inference_step(mixed=True)); the reference's own mixed policy is float16 (Model.py:31-35), see DESIGN.md.
MIXED = False
# mixed emulation only: prenet layer 0 folded into the projection GEMM (what the HIP path does when
# GST_Tacotron.decode_plan(Tv)[1] is true; see `decoder`)
FUSED_PRENET0 = True


def bf16_round(a):
    """float -> nearest-even bfloat16 -> back (finite values), in the input's dtype."""
    a = np.asarray(a)
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32)
    return r.view(np.float32).reshape(a.shape).astype(a.dtype)


def mm(a, b):
    return bf16_round(a) @ bf16_round(b) if MIXED else a @ b


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def same_pad(n_in, k, s):
    """TF padding='same' (SURVEY Appendix A.3, F10): out=ceil(in/s),
    total=max((out-1)*s+k-in,0), before=total//2, after=total-before."""
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    before = total // 2
    return out, before, total - before


def batch_norm(x, w, prefix):
    """Inference BatchNormalization on the last axis (Appendix A.4)."""
    g, b = w[prefix + ".gamma"], w[prefix + ".beta"]
    m, v = w[prefix + ".moving_mean"], w[prefix + ".moving_variance"]
    return g * (x - m) / np.sqrt(v + x.dtype.type(BN_EPS)) + b


def conv1d_same(x, kernel):
    """Conv1D, stride 1, padding same, no bias; x [B,T,C], kernel [k,Cin,Cout]
    (reference Taco2.py:27-33, 137-143; cross-correlation)."""
    k = kernel.shape[0]
    B, T, _ = x.shape
    _, pb, pa = same_pad(T, k, 1)
    xp = np.pad(x, ((0, 0), (pb, pa), (0, 0)))
    y = np.zeros((B, T, kernel.shape[2]), dtype=x.dtype)
    for j in range(k):
        y += mm(xp[:, j:j + T, :], kernel[j])
    return y


def conv2d_same(x, kernel, stride):
    """Conv2D padding same (asymmetric, F10), no bias; x [B,H,W,C] NHWC with
    H=time, W=freq; kernel [kh,kw,Cin,Cout] (reference GST.py:23-29)."""
    kh, kw = kernel.shape[:2]
    B, H, W, _ = x.shape
    Ho, hb, ha = same_pad(H, kh, stride)
    Wo, wb, wa = same_pad(W, kw, stride)
    xp = np.pad(x, ((0, 0), (hb, ha), (wb, wa), (0, 0)))
    y = np.zeros((B, Ho, Wo, kernel.shape[3]), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            y += patch @ kernel[i, j]
    return y


def lstm_cell(x, h, c, kernel, rec, bias):
    """Keras LSTMCell (Appendix A.6): z=x.W+h.U+b, split i,f,c~,o."""
    z = mm(x, kernel) + mm(h, rec) + bias
    u = h.shape[-1]
    i, f = sigmoid(z[:, :u]), sigmoid(z[:, u:2 * u])
    g, o = np.tanh(z[:, 2 * u:3 * u]), sigmoid(z[:, 3 * u:])
    c2 = f * c + i * g
    return o * np.tanh(c2), c2


def lstm_sequence(x, kernel, rec, bias, reverse=False, lengths=None):
    """``lengths`` (masked-mode EXTENSION; the reference has no masks, F5): steps t >= lengths[b] do
    not exist for utterance b -- state stays zero / frozen and the output row is zero, so the backward direction
    starts at lengths[b]-1 with a zero state exactly as if the utterance were run alone."""
    B, T, _ = x.shape
    u = rec.shape[0]
    h = np.zeros((B, u), x.dtype)
    c = np.zeros((B, u), x.dtype)
    out = np.zeros((B, T, u), x.dtype)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        h2, c2 = lstm_cell(x[:, t], h, c, kernel, rec, bias)
        if lengths is not None:
            act = (t < np.asarray(lengths))[:, None]
            h2 = np.where(act, h2, 0.0).astype(x.dtype)
            c2 = np.where(act, c2, c).astype(x.dtype)
        h, c = h2, c2
        out[:, t] = h       # backward outputs are stored back in forward time order (A.7)
    return out


def gru_sequence(x, kernel, rec, bias):
    """Keras GRU, TF2 default reset_after=True (Appendix A.8): order z,r,h;
    bias[0] input bias, bias[1] recurrent bias."""
    B, T, _ = x.shape
    u = rec.shape[0]
    h = np.zeros((B, u), x.dtype)
    out = np.zeros((B, T, u), x.dtype)
    for t in range(T):
        mx = x[:, t] @ kernel + bias[0]
        mh = h @ rec + bias[1]
        z = sigmoid(mx[:, :u] + mh[:, :u])
        r = sigmoid(mx[:, u:2 * u] + mh[:, u:2 * u])
        hh = np.tanh(mx[:, 2 * u:] + r * mh[:, 2 * u:])
        h = z * h + (1.0 - z) * hh
        out[:, t] = h
    return out


def cast_weights(weights, dt):
    return {k: np.asarray(v, dtype=dt) for k, v in weights.items()}


# ----------------------------------------------------------------------------- modules
def encoder(hp, w, tokens, dt, token_lengths=None):
    """Reference Modules/Taco2.py:12-51: Embedding -> 3x(Conv1D, BN, ReLU,
    Dropout=identity at inference) -> Bidirectional LSTM, concat [fwd, bwd].
    ``token_lengths`` switches on the masked-mode extension (A12): every conv sees zeros at t >= length (what
    'same' padding would supply if the utterance were alone) and the BiLSTM covers [0, length) only."""
    x = w["encoder.embedding"][np.asarray(tokens)]
    m = None
    if token_lengths is not None:
        m = (np.arange(x.shape[1])[None, :] < np.asarray(token_lengths)[:, None])[..., None].astype(x.dtype)
    n_conv = len(hp["Tacotron2"]["Encoder"]["Conv"]["Filters"])
    for i in range(n_conv):
        if m is not None:
            x = x * m
        x = conv1d_same(x, w[f"encoder.conv{i}.kernel"])
        x = np.maximum(batch_norm(x, w, f"encoder.conv{i}.bn"), 0)
    p = "encoder.bilstm."
    fwd = lstm_sequence(x, w[p + "fwd.kernel"], w[p + "fwd.recurrent_kernel"], w[p + "fwd.bias"], lengths=token_lengths)
    bwd = lstm_sequence(x, w[p + "bwd.kernel"], w[p + "bwd.recurrent_kernel"], w[p + "bwd.bias"], reverse=True,
                        lengths=token_lengths)
    return np.concatenate([fwd, bwd], axis=-1).astype(dt)


def reference_encoder(hp, w, mels, mel_lengths, dt):
    """Reference Modules/GST.py:12-70.  ``mels`` [B,T_ref,mel] (frame 0 already dropped)."""
    ref = hp["GST"]["Reference_Encoder"]
    x = np.asarray(mels, dt)[..., None]                       # GST.py:54
    for i, s in enumerate(ref["Conv"]["Strides"]):
        x = conv2d_same(x, w[f"gst.ref.conv{i}.kernel"], int(s))
        x = np.maximum(batch_norm(x, w, f"gst.ref.conv{i}.bn"), 0)
    B, T2, F2, C2 = x.shape
    x = x.reshape(B, T2, F2 * C2)                             # GST.py:59-62 (index f*C+c)
    x = gru_sequence(x, w["gst.ref.gru.kernel"], w["gst.ref.gru.recurrent_kernel"], w["gst.ref.gru.bias"])
    prod = int(np.prod(ref["Conv"]["Strides"]))
    idx = np.ceil(np.asarray(mel_lengths, np.int32) / prod).astype(np.int32) - 1   # GST.py:38-40,65-68
    x = x[np.arange(B), idx]
    return np.tanh(x @ w["gst.ref.dense.kernel"] + w["gst.ref.dense.bias"])      # GST.py:42-45,70


def layer_norm(x, gamma, beta, eps=1e-8):
    """Reference Layers.py:254-285: population variance, eps inside the sqrt."""
    mean = x.mean(-1, keepdims=True)
    var = ((x - mean) ** 2).mean(-1, keepdims=True)
    return gamma * ((x - mean) / (var + x.dtype.type(eps)) ** 0.5) + beta


def softmax(x):
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def style_token_layer(hp, w, mels_for_gst, mel_lengths, dt):
    """Reference Modules/GST.py:91-109 + Layers.py:172-214: drop frame 0, reference
    encoder, 4-head unscaled dot-product attention over tanh(tokens), LN(out+q)."""
    ref = reference_encoder(hp, w, np.asarray(mels_for_gst, dt)[:, 1:], mel_lengths, dt)   # GST.py:98
    tokens = np.tanh(w["gst.tokens"])                                        # GST.py:101
    q = ref @ w["gst.mha.query.kernel"] + w["gst.mha.query.bias"]            # [B,A]   Layers.py:174
    v = tokens @ w["gst.mha.value.kernel"] + w["gst.mha.value.bias"]         # [N,A]   Layers.py:175 (key=value :176)
    heads = int(hp["GST"]["Style_Token"]["Attention"]["Head"])
    dh = q.shape[-1] // heads
    out = np.zeros_like(q)
    for h in range(heads):
        sl = slice(h * dh, (h + 1) * dh)
        scores = q[:, sl] @ v[:, sl].T                                      # Layers.py:224, no scale (F13)
        out[:, sl] = softmax(scores) @ v[:, sl]                             # Layers.py:235-237
    return layer_norm(out + q, w["gst.mha.ln.gamma"], w["gst.mha.ln.beta"])  # Layers.py:211


def gst_concat(enc, gst):
    """Reference Modules/GST.py:111-124: memory = [tile(gst) | enc]."""
    B, T, _ = enc.shape
    return np.concatenate([np.broadcast_to(gst[:, None, :], (B, T, gst.shape[-1])), enc], axis=-1)


def prenet(hp, w, x, masks, z0=None):
    """Reference Taco2.py:262-283: Dense relu + Dropout that is ALWAYS on (F3).
    ``masks``: list of keep-masks (1 keep / 0 drop), or None for rate 0.
    ``z0``: layer 0's pre-activations computed elsewhere (mixed-precision emulation of the HIP path's fused form, see
    ``decoder``)."""
    rate = float(hp["Tacotron2"]["Decoder"]["Prenet"]["Dropout_Rate"])
    for i in range(len(hp["Tacotron2"]["Decoder"]["Prenet"]["Size"])):
        pre = z0 if (i == 0 and z0 is not None) else x @ w[f"decoder.prenet{i}.kernel"] + w[f"decoder.prenet{i}.bias"]
        x = np.maximum(pre, 0)
        if rate > 0.0:
            x = x * x.dtype.type(1.0 / (1.0 - rate)) * masks[i]
    return x


def monotonic_alignment(att_type, score, prev, lengths=None):
    """SMA: reference Steps.py:222-229; BMA: Steps.py:171-180,183-199.
    ``lengths`` (masked-mode extension A12): positions >= lengths[b] do not exist for utterance b."""
    if lengths is not None:
        out = np.zeros_like(prev)
        for b, n in enumerate(np.asarray(lengths)):
            out[b:b + 1, :n] = monotonic_alignment(att_type, score[b:b + 1, :n], prev[b:b + 1, :n])
        return out
    p = sigmoid(score)
    if att_type == "SMA":
        shifted = np.zeros_like(prev)
        shifted[:, 1:] = prev[:, :-1] * (1.0 - p[:, :-1])
        return prev * p + shifted
    tiny = np.finfo(np.float32).tiny        # Steps.py:198 with a float32 model dtype
    logs = np.log(np.clip(1.0 - p, tiny, 1.0))
    excl = np.cumsum(logs, axis=-1) - logs                                   # exclusive cumsum
    cp = np.exp(excl)
    return p * cp * np.cumsum(prev / np.clip(cp, 1e-10, 1.0), axis=-1)


def lsa_step(hp, w, query_in, processed_memory, state, lengths=None):
    """EXTENSION: one step of LocationSensitiveAttention restated from the whole-sequence layer
    reference Modules/Attention/Layers.py:345-424 (the reference decoder cannot select it, Taco2.py:66-75).
    ``state`` is what Layers.py:361 feeds the location conv: the SUM of all previous alignments when
    cumulate_weights (default True, zeros before the first step, :356) or the last alignment otherwise."""
    att = hp["Tacotron2"]["Decoder"]["Attention"]
    q = query_in @ w["decoder.attention.query.kernel"] + w["decoder.attention.query.bias"]       # :351
    loc = conv1d_same(state[:, :, None], w["decoder.attention.location_conv.kernel"]) \
        + w["decoder.attention.location_conv.bias"]                                               # :362-363 (Conv1D has a bias)
    loc = loc @ w["decoder.attention.location_dense.kernel"] + w["decoder.attention.location_dense.bias"]   # :364
    score = np.tanh(q[:, None, :] + processed_memory + loc + w["decoder.attention.bias"]).sum(-1)  # :407 (scale = 1)
    if lengths is not None:
        score = np.where(np.arange(score.shape[1])[None, :] < np.asarray(lengths)[:, None], score, -np.inf)
    if att.get("Smoothing", False):
        sg = sigmoid(score)                                                                       # :426-444
        align = sg / sg.sum(-1, keepdims=True)
    else:
        align = softmax(score)                                                                    # :419-420
    ctx = np.einsum("bt,bta->ba", align, processed_memory)                                        # :421
    new_state = state + align if att.get("Cumulate_Weights", True) else align
    return ctx, align, new_state


def attention_step(hp, w, query_in, processed_memory, prev_align, noise, lengths=None):
    """Reference Steps.py:107-166.  ``processed_memory`` is Dense_Value(memory)
    (loop-invariant, F7); the context is a weighted sum of the PROJECTED memory."""
    att = hp["Tacotron2"]["Decoder"]["Attention"]
    q = query_in @ w["decoder.attention.query.kernel"] + w["decoder.attention.query.bias"]     # :122
    score = (w["decoder.attention.v"] * np.tanh(q[:, None, :] + processed_memory)).sum(-1) \
        + w["decoder.attention.score_bias"]                                                     # :152
    sn = att.get("Sigmoid_Noise", 2.0 if att["Type"] == "SMA" else 0.0)
    if sn > 0.0:
        score = score + score.dtype.type(sn) * noise                                            # :169-170 / :220-221
    align = monotonic_alignment(att["Type"], score, prev_align, lengths)
    ctx = np.einsum("bt,bta->ba", align, processed_memory)                                      # :164
    return ctx, align


def process_memory(w, memory):
    return mm(memory, w["decoder.attention.value.kernel"]) + w["decoder.attention.value.bias"]  # Steps.py:123


def decoder(hp, w, memory, dt, prenet_masks=None, attn_noise=None, steps=None, return_states=False, token_lengths=None):
    """Reference Taco2.py:153-228 (training=False branch).
    prenet_masks [steps, n_prenet, B, size] keep-masks; attn_noise [steps, B, T_v] ~ N(0,1)."""
    mel, r = int(hp["Sound"]["Mel_Dim"]), int(hp["Step_Reduction"])
    if steps is None:
        steps = int(hp["Max_Step"]) // r                                     # Taco2.py:213
    B, Tv, _ = memory.shape
    pm = process_memory(w, memory)
    sizes = hp["Tacotron2"]["Decoder"]["RNN"]["Size"]
    hs = [np.zeros((B, s), dt) for s in sizes]
    cs = [np.zeros((B, s), dt) for s in sizes]
    frame = np.zeros((B, mel), dt)                                           # Taco2.py:162-165
    is_lsa = hp["Tacotron2"]["Decoder"]["Attention"]["Type"] == "LSA"
    align = np.zeros((B, Tv), dt)
    if not is_lsa:
        align[:, 0] = 1.0                                                    # Steps.py:201-206
    lsa_state = np.zeros((B, Tv), dt)                                        # Layers.py:356
    pre = np.zeros((B, steps * r, mel), dt)
    stops = np.zeros((B, steps), dt)
    aligns = np.zeros((B, steps, Tv), dt)
    # Mixed-precision emulation only: from step 1 on the HIP path gets prenet layer 0's pre-activations from the
    # projection launch of the previous step -- both layers are linear, frame.W0 + b0 = [h2|ctx].(Wp_last.W0) +
    # (bp_last.W0 + b0) -- with the composed matrix (formed in float64, stored as float32) rounded to bf16 like every
    # other GEMM weight of that mode.  In fp32 the two forms agree to rounding and the oracle keeps the reference's.
    z0 = None
    fuse = MIXED and FUSED_PRENET0 and not is_lsa and len(hp["Tacotron2"]["Decoder"]["Prenet"]["Size"]) == 2
    if fuse:
        last = (r - 1) * mel
        W0 = np.asarray(w["decoder.prenet0.kernel"], np.float64)
        Wp = np.asarray(w["decoder.projection.kernel"], np.float64)[:, last:last + mel]
        bp = np.asarray(w["decoder.projection.bias"], np.float64)[last:last + mel]
        Wz = (Wp @ W0).astype(np.float32).astype(dt)
        bz = (np.asarray(w["decoder.prenet0.bias"], np.float64) + bp @ W0).astype(np.float32).astype(dt)
    for t in range(steps):
        masks = None if prenet_masks is None else prenet_masks[t]
        p = prenet(hp, w, frame, masks, z0)                                  # Taco2.py:106
        noise = None if attn_noise is None else attn_noise[t]
        if is_lsa:
            if token_lengths is not None:       # processed memory beyond the length is irrelevant (align = 0 there)
                lsa_in = lsa_state * (np.arange(Tv)[None, :] < np.asarray(token_lengths)[:, None])
            else:
                lsa_in = lsa_state
            ctx, align, lsa_state = lsa_step(hp, w, p, pm, lsa_in, token_lengths)
        else:
            ctx, align = attention_step(hp, w, p, pm, align, noise, token_lengths)   # Taco2.py:107-109
        x = np.concatenate([p, ctx], -1)                                     # :110
        for i in range(len(sizes)):                                          # :111 StackedRNNCells
            hs[i], cs[i] = lstm_cell(x, hs[i], cs[i], w[f"decoder.lstm{i}.kernel"],
                                     w[f"decoder.lstm{i}.recurrent_kernel"], w[f"decoder.lstm{i}.bias"])
            x = hs[i]
        y = mm(np.concatenate([x, ctx], -1), w["decoder.projection.kernel"]) + w["decoder.projection.bias"]  # :112-113
        if fuse:
            z0 = mm(np.concatenate([x, ctx], -1), Wz) + bz
        pre[:, t * r:(t + 1) * r] = y[:, :mel * r].reshape(B, r, mel)        # :194-201
        stops[:, t] = y[:, mel * r]
        aligns[:, t] = align
        frame = pre[:, (t + 1) * r - 1]                                      # :186 decodings[:, -1] (F14)
    if return_states:
        return pre, stops, aligns, (hs, cs)
    return pre, stops, aligns


def postnet(hp, w, pre, dt):
    """Reference Taco2.py:131-149,230: 5x(Conv1D+BN), tanh after layers 0..2 only (F9), + residual."""
    n = len(hp["Tacotron2"]["Decoder"]["Conv"]["Filters"]) + 1
    x = pre
    for i in range(n):
        x = batch_norm(conv1d_same(x, w[f"postnet.conv{i}.kernel"]), w, f"postnet.conv{i}.bn")
        if i < n - 2:
            x = np.tanh(x)
    return x + pre


def conv1d_same_general(x, kernel):
    """Conv1D stride 1 padding same for ANY kernel size (even sizes pad asymmetrically: before = (k-1)//2)."""
    return conv1d_same(x, kernel)       # conv1d_same already uses same_pad(); kept as a named alias for the bank


def maxpool1d_same2(y):
    """MaxPool1D(pool 2, strides 1, 'same') (reference Taco2.py:319-324): TF pads one frame AFTER and padding never
    wins a max, so frame t = max(y[t], y[t+1]) and the last frame is unchanged."""
    return np.concatenate([np.maximum(y[:, :-1], y[:, 1:]), y[:, -1:]], 1)


def highway(y, w_relu, b_relu, w_sig, b_sig):
    """One Highwaynet layer (reference Taco2.py:409-424): H*T + x*(1-T), H = relu dense, T = sigmoid dense."""
    h = np.maximum(mm(y, w_relu) + b_relu, 0)
    t = sigmoid(mm(y, w_sig) + b_sig)
    return h * t + y * (1.0 - t)


def vocoder_taco1(hp, w, mels, dt):
    """Reference Taco2.py:234-260 (Vocoder_Taco1) + CBHG :285-380 + ConvBank :383-407 + Highwaynet :409-424:
    mel [B,T,mel] -> linear spectrogram [B,T,Spectrogram_Dim].  SURVEY row N1."""
    cb = hp["Vocoder_Taco1"]["CBHG"]
    x = np.asarray(mels, dt)
    banks = []
    for i in range(int(cb["Conv_Bank"]["Stack_Count"])):                      # kernel sizes 1..Stack_Count, all on the INPUT
        y = conv1d_same(x, w[f"vocoder.convbank{i}.kernel"])
        banks.append(np.maximum(batch_norm(y, w, f"vocoder.convbank{i}.bn"), 0))
    y = np.concatenate(banks, -1)                                             # :404-407
    y = maxpool1d_same2(y)
    n = len(cb["Conv1D"]["Filters"])
    for i in range(n):                                                        # :326-340
        y = batch_norm(conv1d_same(y, w[f"vocoder.proj{i}.kernel"]), w, f"vocoder.proj{i}.bn")
        if i < n - 1:
            y = np.maximum(y, 0)
    if "vocoder.proj_dense.kernel" in w:                                      # :342-345
        y = mm(y, w["vocoder.proj_dense.kernel"]) + w["vocoder.proj_dense.bias"]
    y = y + x                                                                 # residual :373
    if "vocoder.highway_in.kernel" in w:                                      # :348-351
        y = mm(y, w["vocoder.highway_in.kernel"]) + w["vocoder.highway_in.bias"]
    for i in range(int(cb["Highwaynet"]["Count"])):                           # :409-424
        y = highway(y, w[f"vocoder.highway{i}.relu.kernel"], w[f"vocoder.highway{i}.relu.bias"],
                    w[f"vocoder.highway{i}.sigmoid.kernel"], w[f"vocoder.highway{i}.sigmoid.bias"])
    p = "vocoder.bilstm."
    fwd = lstm_sequence(y, w[p + "fwd.kernel"], w[p + "fwd.recurrent_kernel"], w[p + "fwd.bias"])
    bwd = lstm_sequence(y, w[p + "bwd.kernel"], w[p + "bwd.recurrent_kernel"], w[p + "bwd.bias"], reverse=True)
    y = np.concatenate([fwd, bwd], -1)                                        # :357-361
    return mm(y, w["vocoder.dense.kernel"]) + w["vocoder.dense.bias"]        # :252-260


def inference_step(hp, weights, tokens, mels_for_gst=None, mel_lengths_for_gst=None,
                   prenet_masks=None, attn_noise=None, steps=None, dt=np.float64, token_lengths=None, with_vocoder=False,
                   mixed=False, fused_prenet0=True):
    """Reference Model.py:249-255 with the wiring of Model.py:108-129,145-156.  ``mixed`` emulates the HIP path's
    Use_Mixed_Precision mode (bf16 GEMM operands, see `mm`); ``fused_prenet0`` (mixed only) says whether that path folds
    prenet layer 0 into the projection GEMM for this shape (``GST_Tacotron.decode_plan``).
    Returns (mels [B,S*r,mel] post-net, stops [B,S], spectrograms [B,S*r,Spectrogram_Dim] (``with_vocoder``; else None),
    alignments [B,S,T_v])
    plus a dict of intermediates for per-module parity tests."""
    global MIXED, FUSED_PRENET0
    prev_mixed, MIXED = MIXED, bool(mixed)
    prev_fused, FUSED_PRENET0 = FUSED_PRENET0, bool(fused_prenet0)
    try:
        return _inference_step(hp, weights, tokens, mels_for_gst, mel_lengths_for_gst, prenet_masks, attn_noise, steps, dt,
                               token_lengths, with_vocoder)
    finally:
        MIXED, FUSED_PRENET0 = prev_mixed, prev_fused


def _inference_step(hp, weights, tokens, mels_for_gst, mel_lengths_for_gst, prenet_masks, attn_noise, steps, dt,
                    token_lengths, with_vocoder):
    w = cast_weights(weights, dt)
    enc = encoder(hp, w, tokens, dt, token_lengths)        # token_lengths=None: the reference's unmasked behaviour
    inter = {"encoder": enc}
    memory = enc
    if hp["GST"]["Use"]:
        gst = style_token_layer(hp, w, mels_for_gst, mel_lengths_for_gst, dt)
        inter["gst"] = gst
        memory = gst_concat(enc, gst)
    if prenet_masks is not None:
        prenet_masks = np.asarray(prenet_masks, dt)
    if attn_noise is not None:
        attn_noise = np.asarray(attn_noise, dt)
    pre, stops, aligns = decoder(hp, w, memory, dt, prenet_masks, attn_noise, steps, token_lengths=token_lengths)
    inter["pre_mel"] = pre
    mels = postnet(hp, w, pre, dt)
    spec = vocoder_taco1(hp, w, mels, dt) if with_vocoder else None          # Model.py:126-129 (training=False)
    return mels, stops, spec, aligns, inter

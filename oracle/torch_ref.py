"""CPU ORACLE #2 (test infrastructure, NOT product code) -- an independent,
reference-faithful PyTorch-CPU eager restatement of the TF2 graph.

PARITY UNPINNED (see oracle/oracle_np.py header): TensorFlow cannot be run here.

Two jobs:
  1. cross-check oracle_np.py with a second implementation built from torch's own
     conv / LSTMCell / GRUCell primitives (weights re-laid-out from the Keras layouts);
  2. be the ``cpu_baseline`` ("port") that bench.py times on the GPU box's host cores:
     it keeps the reference's algorithmic choices -- the attention Value projection is
     recomputed EVERY decoder step (reference Steps.py:123), outputs grow by concat every
     step (reference Taco2.py:203-205), one eager op dispatch per TF op.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def _same_pad(n_in, k, s):
    out = -(-n_in // s)
    total = max((out - 1) * s + k - n_in, 0)
    return total // 2, total - total // 2


class TorchReference:
    def __init__(self, hp, weights, dtype=torch.float32):
        self.hp = hp
        self.dt = dtype
        self.w = {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
        w = self.w
        # torch-native layouts (built once, like Keras builds its variables once)
        self.conv1d_w = {}
        for name in w:
            if name.endswith(".kernel") and w[name].dim() == 3:
                self.conv1d_w[name] = w[name].permute(2, 1, 0).contiguous()       # [k,Cin,Cout] -> [Cout,Cin,k]
        self.conv2d_w = {name: w[name].permute(3, 2, 0, 1).contiguous()
                         for name in w if name.endswith(".kernel") and w[name].dim() == 4}
        self.bn = {}
        for name in w:
            if name.endswith(".bn.gamma"):
                p = name[:-len(".gamma")]
                scale = w[p + ".gamma"] / torch.sqrt(w[p + ".moving_variance"] + BN_EPS)
                self.bn[p] = (scale, w[p + ".beta"] - w[p + ".moving_mean"] * scale)

    # ---- primitives -----------------------------------------------------------------
    def _bn(self, x, prefix):
        g, b = self.w[prefix + ".gamma"], self.w[prefix + ".beta"]
        m, v = self.w[prefix + ".moving_mean"], self.w[prefix + ".moving_variance"]
        return F.batch_norm(x, m, v, g, b, training=False, eps=BN_EPS)

    def _conv1d_bn(self, x, prefix):
        """x [B,T,C] channels-last like Keras."""
        wt = self.conv1d_w[prefix + ".kernel"]
        pb, pa = _same_pad(x.shape[1], wt.shape[2], 1)
        y = F.conv1d(F.pad(x.transpose(1, 2), (pb, pa)), wt)
        return self._bn(y, prefix + ".bn").transpose(1, 2)

    def _lstm_cell(self, x, h, c, prefix):
        w = self.w
        z = x @ w[prefix + ".kernel"] + h @ w[prefix + ".recurrent_kernel"] + w[prefix + ".bias"]
        i, f, g, o = torch.chunk(z, 4, dim=-1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c2), c2

    def _lstm_seq_native(self, x, prefix, reverse):
        """torch.nn.functional-free LSTM via torch's fused cell (gate order i,f,g,o == Keras i,f,c,o)."""
        w = self.w
        u = w[prefix + ".recurrent_kernel"].shape[0]
        w_ih = w[prefix + ".kernel"].t().contiguous()
        w_hh = w[prefix + ".recurrent_kernel"].t().contiguous()
        b_ih = w[prefix + ".bias"]
        b_hh = torch.zeros_like(b_ih)
        B, T, _ = x.shape
        h = torch.zeros(B, u, dtype=self.dt)
        c = torch.zeros(B, u, dtype=self.dt)
        outs = [None] * T
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            h, c = torch._VF.lstm_cell(x[:, t], (h, c), w_ih, w_hh, b_ih, b_hh)
            outs[t] = h
        return torch.stack(outs, dim=1)

    # ---- modules ---------------------------------------------------------------------
    def encoder(self, tokens):
        """reference Taco2.py:12-51"""
        x = F.embedding(torch.as_tensor(np.asarray(tokens), dtype=torch.long), self.w["encoder.embedding"])
        for i in range(len(self.hp["Tacotron2"]["Encoder"]["Conv"]["Filters"])):
            x = torch.relu(self._conv1d_bn(x, f"encoder.conv{i}"))
        fwd = self._lstm_seq_native(x, "encoder.bilstm.fwd", False)
        bwd = self._lstm_seq_native(x, "encoder.bilstm.bwd", True)
        return torch.cat([fwd, bwd], dim=-1)

    def reference_encoder(self, mels, mel_lengths):
        """reference GST.py:47-70"""
        ref = self.hp["GST"]["Reference_Encoder"]
        x = mels.unsqueeze(1)                                        # NCHW: [B,1,T,F]
        for i, s in enumerate(ref["Conv"]["Strides"]):
            wt = self.conv2d_w[f"gst.ref.conv{i}.kernel"]
            hb, ha = _same_pad(x.shape[2], wt.shape[2], s)
            wb, wa = _same_pad(x.shape[3], wt.shape[3], s)
            x = F.conv2d(F.pad(x, (wb, wa, hb, ha)), wt, stride=s)
            x = torch.relu(self._bn(x, f"gst.ref.conv{i}.bn"))
        B, C, T2, F2 = x.shape
        x = x.permute(0, 2, 3, 1).reshape(B, T2, F2 * C)             # NHWC reshape: index f*C+c
        # torch GRUCell: gates (r,z,n) vs Keras (z,r,h); both "reset-after"
        w = self.w
        u = w["gst.ref.gru.recurrent_kernel"].shape[0]
        perm = torch.cat([torch.arange(u, 2 * u), torch.arange(0, u), torch.arange(2 * u, 3 * u)])
        w_ih = w["gst.ref.gru.kernel"].t()[perm].contiguous()
        w_hh = w["gst.ref.gru.recurrent_kernel"].t()[perm].contiguous()
        b_ih = w["gst.ref.gru.bias"][0][perm].contiguous()
        b_hh = w["gst.ref.gru.bias"][1][perm].contiguous()
        h = torch.zeros(B, u, dtype=self.dt)
        outs = []
        for t in range(T2):
            h = torch._VF.gru_cell(x[:, t], h, w_ih, w_hh, b_ih, b_hh)
            outs.append(h)
        seq = torch.stack(outs, dim=1)
        prod = int(np.prod(ref["Conv"]["Strides"]))
        idx = torch.as_tensor(np.ceil(np.asarray(mel_lengths) / prod).astype(np.int64) - 1)
        g = seq[torch.arange(B), idx]
        return torch.tanh(g @ w["gst.ref.dense.kernel"] + w["gst.ref.dense.bias"])

    def style_token_layer(self, mels_for_gst, mel_lengths):
        """reference GST.py:91-109, Layers.py:172-214 (split/concat on the batch axis, like the reference)"""
        w = self.w
        mels = torch.as_tensor(np.asarray(mels_for_gst), dtype=self.dt)[:, 1:]
        ref = self.reference_encoder(mels, mel_lengths)
        B = ref.shape[0]
        tokens = torch.tanh(w["gst.tokens"]).unsqueeze(0).repeat(B, 1, 1)
        q = ref.unsqueeze(1) @ w["gst.mha.query.kernel"] + w["gst.mha.query.bias"]
        v = tokens @ w["gst.mha.value.kernel"] + w["gst.mha.value.bias"]
        heads = int(self.hp["GST"]["Style_Token"]["Attention"]["Head"])
        q_s = torch.cat(torch.chunk(q, heads, dim=-1), dim=0)
        v_s = torch.cat(torch.chunk(v, heads, dim=-1), dim=0)
        scores = q_s @ v_s.transpose(1, 2)
        res = torch.softmax(scores, dim=-1) @ v_s
        res = torch.cat(torch.chunk(res, heads, dim=0), dim=-1) + q
        var, mean = torch.var_mean(res, dim=-1, keepdim=True, unbiased=False)
        out = (res - mean) / (var + 1e-8) ** 0.5 * w["gst.mha.ln.gamma"] + w["gst.mha.ln.beta"]
        return out.squeeze(1)

    def decoder_step(self, memory, frame, prev_align, states, masks, noise):
        """reference Taco2.py:96-120 with Steps.py:107-166 -- Value projection NOT hoisted."""
        w, hp = self.w, self.hp
        x = frame
        rate = float(hp["Tacotron2"]["Decoder"]["Prenet"]["Dropout_Rate"])
        for i in range(len(hp["Tacotron2"]["Decoder"]["Prenet"]["Size"])):
            x = torch.relu(x @ w[f"decoder.prenet{i}.kernel"] + w[f"decoder.prenet{i}.bias"])
            if rate > 0.0:
                x = x * (1.0 / (1.0 - rate)) * masks[i]
        att = hp["Tacotron2"]["Decoder"]["Attention"]
        q = x @ w["decoder.attention.query.kernel"] + w["decoder.attention.query.bias"]
        value = memory @ w["decoder.attention.value.kernel"] + w["decoder.attention.value.bias"]   # every step (F7)
        if att["Type"] == "LSA":        # extension A13, restated from Layers.py:345-424 with torch's own conv1d
            state = prev_align          # cumulative (or last) alignment
            k = w["decoder.attention.location_conv.kernel"]
            pb, pa = _same_pad(state.shape[1], k.shape[0], 1)
            loc = F.conv1d(F.pad(state.unsqueeze(1), (pb, pa)), k.permute(2, 1, 0).contiguous(),
                           w["decoder.attention.location_conv.bias"]).transpose(1, 2)
            loc = loc @ w["decoder.attention.location_dense.kernel"] + w["decoder.attention.location_dense.bias"]
            score = torch.tanh(q.unsqueeze(1) + value + loc + w["decoder.attention.bias"]).sum(-1)
            if att.get("Smoothing", False):
                sg = torch.sigmoid(score)
                align = sg / sg.sum(-1, keepdim=True)
            else:
                align = torch.softmax(score, dim=-1)
            ctx = (align.unsqueeze(1) @ value).squeeze(1)
            new_state = state + align if att.get("Cumulate_Weights", True) else align
            y = torch.cat([x, ctx], dim=-1)
            new_states = []
            for i in range(len(hp["Tacotron2"]["Decoder"]["RNN"]["Size"])):
                h, c = self._lstm_cell(y, states[i][0], states[i][1], f"decoder.lstm{i}")
                new_states.append((h, c))
                y = h
            y = torch.cat([y, ctx], dim=-1) @ w["decoder.projection.kernel"] + w["decoder.projection.bias"]
            return y[:, :-1], y[:, -1:], (align, new_state), new_states
        score = torch.sum(w["decoder.attention.v"] * torch.tanh(q.unsqueeze(1) + value), dim=-1) \
            + w["decoder.attention.score_bias"]
        sn = att.get("Sigmoid_Noise", 2.0 if att["Type"] == "SMA" else 0.0)
        if sn > 0.0:
            score = score + sn * noise
        p = torch.sigmoid(score)
        if att["Type"] == "SMA":
            pad = torch.zeros(p.shape[0], 1, dtype=self.dt)
            align = prev_align * p + torch.cat([pad, prev_align[:, :-1] * (1.0 - p[:, :-1])], dim=-1)
        else:
            tiny = float(np.finfo(np.float32).tiny)
            logs = torch.log(torch.clamp(1.0 - p, tiny, 1.0))
            cp = torch.exp(torch.cumsum(logs, dim=-1) - logs)
            align = p * cp * torch.cumsum(prev_align / torch.clamp(cp, 1e-10, 1.0), dim=-1)
        ctx = (align.unsqueeze(1) @ value).squeeze(1)
        y = torch.cat([x, ctx], dim=-1)
        new_states = []
        for i in range(len(hp["Tacotron2"]["Decoder"]["RNN"]["Size"])):
            h, c = self._lstm_cell(y, states[i][0], states[i][1], f"decoder.lstm{i}")
            new_states.append((h, c))
            y = h
        y = torch.cat([y, ctx], dim=-1) @ w["decoder.projection.kernel"] + w["decoder.projection.bias"]
        return y[:, :-1], y[:, -1:], align, new_states

    def decoder(self, memory, prenet_masks, attn_noise, steps=None):
        """reference Taco2.py:153-232 -- concat-grown loop state like the reference (F8)."""
        hp = self.hp
        mel, r = int(hp["Sound"]["Mel_Dim"]), int(hp["Step_Reduction"])
        if steps is None:
            steps = int(hp["Max_Step"]) // r
        B, Tv, _ = memory.shape
        decodings = torch.zeros(B, 1, mel, dtype=self.dt)
        stops = torch.zeros(B, 0, dtype=self.dt)
        is_lsa = hp["Tacotron2"]["Decoder"]["Attention"]["Type"] == "LSA"
        alignments = F.one_hot(torch.zeros(B, dtype=torch.long), Tv).to(self.dt).unsqueeze(1)
        lsa_state = torch.zeros(B, Tv, dtype=self.dt)
        states = [(torch.zeros(B, s, dtype=self.dt), torch.zeros(B, s, dtype=self.dt))
                  for s in hp["Tacotron2"]["Decoder"]["RNN"]["Size"]]
        for t in range(steps):
            masks = None if prenet_masks is None else prenet_masks[t]
            noise = None if attn_noise is None else attn_noise[t]
            if is_lsa:
                dec, stop, (align, lsa_state), states = self.decoder_step(memory, decodings[:, -1], lsa_state, states, masks, noise)
            else:
                dec, stop, align, states = self.decoder_step(memory, decodings[:, -1], alignments[:, -1], states, masks, noise)
            decodings = torch.cat([decodings, dec.reshape(B, r, mel)], dim=1)
            stops = torch.cat([stops, stop], dim=-1)
            alignments = torch.cat([alignments, align.unsqueeze(1)], dim=1)
        decodings = decodings[:, 1:]
        alignments = alignments[:, 1:]
        x = decodings
        n = len(hp["Tacotron2"]["Decoder"]["Conv"]["Filters"]) + 1
        for i in range(n):
            x = self._conv1d_bn(x, f"postnet.conv{i}")
            if i < n - 2:
                x = torch.tanh(x)
        return decodings, x + decodings, stops, alignments

    def vocoder(self, mels):
        """reference Taco2.py:234-260, 285-424 with torch's own conv1d / max_pool1d / fused LSTM cell"""
        w, cb = self.w, self.hp["Vocoder_Taco1"]["CBHG"]
        x = mels
        banks = []
        for i in range(int(cb["Conv_Bank"]["Stack_Count"])):
            banks.append(torch.relu(self._conv1d_bn(x, f"vocoder.convbank{i}")))
        y = torch.cat(banks, dim=-1)
        y = F.max_pool1d(F.pad(y.transpose(1, 2), (0, 1), value=float("-inf")), kernel_size=2, stride=1).transpose(1, 2)
        n = len(cb["Conv1D"]["Filters"])
        for i in range(n):
            y = self._conv1d_bn(y, f"vocoder.proj{i}")
            if i < n - 1:
                y = torch.relu(y)
        if "vocoder.proj_dense.kernel" in w:
            y = y @ w["vocoder.proj_dense.kernel"] + w["vocoder.proj_dense.bias"]
        y = y + x
        if "vocoder.highway_in.kernel" in w:
            y = y @ w["vocoder.highway_in.kernel"] + w["vocoder.highway_in.bias"]
        for i in range(int(cb["Highwaynet"]["Count"])):
            h = torch.relu(y @ w[f"vocoder.highway{i}.relu.kernel"] + w[f"vocoder.highway{i}.relu.bias"])
            t = torch.sigmoid(y @ w[f"vocoder.highway{i}.sigmoid.kernel"] + w[f"vocoder.highway{i}.sigmoid.bias"])
            y = h * t + y * (1.0 - t)
        fwd = self._lstm_seq_native(y, "vocoder.bilstm.fwd", False)
        bwd = self._lstm_seq_native(y, "vocoder.bilstm.bwd", True)
        return torch.cat([fwd, bwd], dim=-1) @ w["vocoder.dense.kernel"] + w["vocoder.dense.bias"]

    def inference_step(self, tokens, mels_for_gst=None, mel_lengths_for_gst=None,
                       prenet_masks=None, attn_noise=None, steps=None, with_vocoder=False):
        """reference Model.py:249-255 (vocoder excluded; north_star metric is mel frames)."""
        with torch.no_grad():
            enc = self.encoder(tokens)
            memory = enc
            if self.hp["GST"]["Use"]:
                gst = self.style_token_layer(mels_for_gst, mel_lengths_for_gst)
                memory = torch.cat([gst.unsqueeze(1).repeat(1, enc.shape[1], 1), enc], dim=-1)
            if prenet_masks is not None:
                prenet_masks = torch.as_tensor(np.asarray(prenet_masks), dtype=self.dt)
            if attn_noise is not None:
                attn_noise = torch.as_tensor(np.asarray(attn_noise), dtype=self.dt)
            pre, mels, stops, aligns = self.decoder(memory, prenet_masks, attn_noise, steps)
            spec = self.vocoder(mels) if with_vocoder else None
        return mels, stops, spec, aligns, {"pre_mel": pre, "encoder": enc}

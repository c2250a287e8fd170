"""``GST_Tacotron`` -- host-side mirror of the reference class of the same name
(reference Model.py:37-459), inference methods only:

    GST_Tacotron(is_Training=False).Restore(path)
    .Inference_Step(tokens, token_lengths, initial_mels, mels_for_gst, mel_lengths_for_gst)
    .Inference_GST_Step(mels_for_gst, mel_lengths_for_gst)
    .Inference(sentence_List, mel_List_for_GST)

Same names, argument order/meaning and error behaviour; the Keras functional model
behind them (Model.py:145-156) is replaced by the HIP kernels behind include/gsttaco.h.
PyTorch is used for device memory and streams only.  Differences, all additive:
  * the config is passed explicitly instead of being read from the CWD at import time;
  * randomness the reference draws unseeded (prenet dropout that is live at inference,
    Taco2.py:283; SMA sigmoid noise, Steps.py:220-221) can be injected (``prenet_masks``,
    ``attn_noise``) for parity runs, otherwise it is generated on the GPU from ``seed``;
  * ``token_lengths`` / ``initial_mels`` are accepted and ignored exactly like the
    reference ignores them at inference (Model.py:249-253, Taco2.py:161);
  * the CBHG vocoder output (3rd element of the returned tuple, Vocoder_Taco1) is computed when
    ``with_vocoder=True`` (gsttaco_vocoder); by default it is None, because the north-star metric
    (mel frames) excludes it.
"""
import ctypes
import os
from datetime import datetime

import numpy as np
import torch

from . import capi, weights as weights_mod
from .feeder import Feeder
from .hparams import Dims, load_hp, load_token_dict


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class GST_Tacotron:
    def __init__(self, is_Training=False, hyper_parameters=None, device=None,
                 max_batch=32, max_tokens=256, max_ref_frames=1025, max_wav_seconds=20.0):
        if is_Training:
            raise NotImplementedError("only the inference hot path is implemented (training is out of scope)")
        self.hp_Dict = load_hp(hyper_parameters)
        self.token_Index_Dict = load_token_dict(self.hp_Dict)
        self.dims = Dims(self.hp_Dict, vocab=len(self.token_Index_Dict))
        self.feeder = Feeder(self.hp_Dict, self.token_Index_Dict, mel_frontend=self.Mel_Generate)
        if device is None:
            dev = self.hp_Dict.get("Device", "0")
            device = int(dev) if str(dev).lstrip("-").isdigit() and int(dev) >= 0 else 0
        self.device_index = int(device)
        self.ctx = capi.Context(self.hp_Dict, vocab=len(self.token_Index_Dict), device=self.device_index,
                                max_batch=max_batch, max_tokens=max_tokens, max_ref_frames=max_ref_frames,
                                max_wav_seconds=max_wav_seconds if self.dims.audio else 0.0)
        self._ready = False
        self.seed = 0

    # ------------------------------------------------------------------ weights
    def Restore(self, checkpoint_File_Path=None, weights=None):
        """reference Model.py:267-276.  Loads (a) a flat ``.npz`` weight file (names/shapes:
        gst_tacotron_amd.weights.manifest), (b) a reference ``tf.train.Checkpoint`` prefix (``<prefix>.index`` +
        ``.data-*``, read without TensorFlow by gst_tacotron_amd.tf_checkpoint -- SURVEY N3), (c) with no argument the
        latest checkpoint under ``Checkpoint_Path`` like the reference, or (d) an in-memory dict; prints and returns
        like the reference when nothing is found."""
        from . import tf_checkpoint
        if weights is None:
            if checkpoint_File_Path is None:
                checkpoint_File_Path = tf_checkpoint.latest_checkpoint(self.hp_Dict.get("Checkpoint_Path", "."))   # :268-270
            if checkpoint_File_Path is not None and os.path.exists(str(checkpoint_File_Path) + ".index"):
                # verify=True: every tensor's crc32c is checked (native gsttaco_crc32c), so a truncated or corrupted
                # .data shard fails here instead of loading silently
                weights = tf_checkpoint.load_reference_checkpoint(checkpoint_File_Path, self.hp_Dict,
                                                                  vocab=len(self.token_Index_Dict), verify=True)
            elif checkpoint_File_Path is not None and os.path.exists(checkpoint_File_Path):
                weights = weights_mod.load_npz(checkpoint_File_Path)
            else:
                print("There is no checkpoint.")
                return self
        weights_mod.check_weights(self.hp_Dict, weights, vocab=len(self.token_Index_Dict))
        if not torch.cuda.is_available():
            raise capi.GstTacoError(-2, "no HIP device: the gfx950 kernels are the only compute path (no CPU fallback)")
        self.ctx.load_weights(weights)
        self.ctx.finalize()
        self._ready = True
        if checkpoint_File_Path is not None:
            print("Checkpoint '{}' is loaded.".format(checkpoint_File_Path))       # reference Model.py:276, after the load
        return self

    # ------------------------------------------------------------------ helpers
    @property
    def device(self):
        return torch.device("cuda", self.device_index)

    def _dev(self, a, dtype):
        if a is None:
            return None
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _require_ready(self):
        if not self._ready:
            raise capi.GstTacoError(-4, "weights not loaded (call Restore first)")

    # ------------------------------------------------------------------ hot path
    def Inference_Step(self, tokens, token_lengths=None, initial_mels=None, mels_for_gst=None,
                       mel_lengths_for_gst=None, prenet_masks=None, attn_noise=None, seed=None,
                       steps=None, return_pre_mel=False, masked=False, with_vocoder=False):
        """reference Model.py:249-255.  Returns (mel_Logits [B,S*r,mel], stop_Logits [B,S],
        spectrogram_Logits ([B,S*r,Spectrogram_Dim] with ``with_vocoder=True``, else None), alignments [B,S,T_v]) as
        CUDA tensors on the current stream.
        ``masked=True`` (extension): honour ``token_lengths`` so each utterance of a ragged batch equals
        that utterance run alone; the default ignores them like the reference does.
        ``with_vocoder=True`` also runs Vocoder_Taco1 (CBHG) and returns spectrogram_Logits [B,S*r,513]
        as the third element like the reference; the default returns None there (the north-star metric excludes it)."""
        self._require_ready()
        d = self.dims
        tok = self._dev(tokens, torch.int32)
        if tok.dim() != 2:
            raise ValueError("tokens must be [batch, time]")
        B, Tv = tok.shape
        tlen = None
        if masked:
            if token_lengths is None:
                raise ValueError("masked=True needs token_lengths")
            tlen = self._dev(token_lengths, torch.int32)
            if tuple(tlen.shape) != (B,):
                raise ValueError("token_lengths must be [batch]")
        mels = lens = None
        Tref1 = 0
        if d.gst:
            if mels_for_gst is None or mel_lengths_for_gst is None:
                raise ValueError("GST is enabled, but no mel information.")
            mels = self._dev(mels_for_gst, torch.float32)
            lens = self._dev(mel_lengths_for_gst, torch.int32)
            if mels.dim() != 3 or mels.shape[0] != B or mels.shape[2] != d.mel or lens.shape != (B,):
                raise ValueError("mels_for_gst must be [batch, frames+1, Mel_Dim] with mel_lengths_for_gst [batch]")
            Tref1 = mels.shape[1]
        S = d.steps if steps is None else int(steps)
        masks = self._dev(prenet_masks, torch.float32)
        noise = self._dev(attn_noise, torch.float32)
        if masks is not None and masks.numel() != S * B * sum(d.prenet):
            raise ValueError("prenet_masks must be [steps, 2, batch, prenet]")
        if noise is not None and tuple(noise.shape) != (S, B, Tv):
            raise ValueError("attn_noise must be [steps, batch, T_v]")
        mel = torch.empty((B, S * d.r, d.mel), dtype=torch.float32, device=self.device)
        pre = torch.empty_like(mel) if return_pre_mel else None
        spec = None
        if with_vocoder:
            if not d.vocoder:
                raise ValueError("Hyper_Parameters has no Vocoder_Taco1 section")
            spec = torch.empty((B, S * d.r, d.spec), dtype=torch.float32, device=self.device)
        stop = torch.empty((B, S), dtype=torch.float32, device=self.device)
        align = torch.empty((B, S, Tv), dtype=torch.float32, device=self.device)
        if seed is None:
            self.seed += 1
            seed = self.seed
        with torch.cuda.device(self.device):
            self.ctx.check(self.ctx.lib.gsttaco_inference_step(
                self.ctx.handle, _ptr(tok), _ptr(tlen), _ptr(mels), _ptr(lens), _ptr(masks), _ptr(noise),
                ctypes.c_uint64(int(seed)), B, Tv, Tref1, S, _ptr(mel), _ptr(stop), _ptr(align), _ptr(pre), _ptr(spec),
                self._stream()))
        if return_pre_mel:
            return mel, stop, spec, align, pre
        return mel, stop, spec, align

    def Inference_GST(self, wav_List, tag_List=None, label=None):
        """reference Model.py:427-446: style embeddings [B, Attention.Size] of the wavs; with ``tag_List`` the table
        of Model.py:448-459 is written too."""
        if not self.hp_Dict["GST"]["Use"]:
            raise NotImplementedError("GST is not used")
        print("GST Inference running...")
        gsts = self.Inference_GST_Step(**self.feeder.Get_Inference_GST_Pattern(wav_List))
        if tag_List is not None:
            self.Export_GST(wav_List, tag_List, gsts, label or datetime.now().strftime("%Y%m%d.%H%M%S"))
        return gsts

    def Inference_GST_Step(self, mels_for_gst, mel_lengths_for_gst):
        """reference Model.py:257-265"""
        if not self.hp_Dict["GST"]["Use"]:
            raise NotImplementedError("GST is not used")
        self._require_ready()
        mels = self._dev(mels_for_gst, torch.float32)
        lens = self._dev(mel_lengths_for_gst, torch.int32)
        B, Tref1 = mels.shape[0], mels.shape[1]
        gst = torch.empty((B, self.dims.gst_att), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self.ctx.check(self.ctx.lib.gsttaco_gst(self.ctx.handle, _ptr(mels), _ptr(lens), B, Tref1, _ptr(gst), self._stream()))
        return gst

    def Mel_Generate(self, wav_List, top_db=60):
        """Batched reference Pattern_Generator.Mel_Generate(path, top_db, range_Ignore=True) (Pattern_Generator.py:39-60)
        on the GPU, already in the mels_for_gst layout of Feeder.py:204-225: returns (mels_for_gst [B, 1+max_len, Mel_Dim]
        with a zero frame 0 and zero padding, mel_lengths_for_gst [B]) as device tensors.  ``wav_List`` holds wav paths
        or 1-D float sample arrays at Sound.Sample_Rate.  Works before Restore (no weights involved)."""
        from .audio import as_signal
        d = self.dims
        if not d.audio or not self.ctx.cfg.max_wav_samples:
            raise ValueError("the audio front end needs the Sound section of Hyper_Parameters and max_wav_seconds > 0")
        if not torch.cuda.is_available():
            raise capi.GstTacoError(-2, "no HIP device: the gfx950 kernels are the only compute path (no CPU fallback)")
        sigs = [as_signal(w, d.sample_rate) for w in wav_List]
        B, ld = len(sigs), max(s.shape[0] for s in sigs)
        host = np.zeros((B, ld), dtype=np.float32)
        for i, s_ in enumerate(sigs):
            host[i, :s_.shape[0]] = s_
        wav = torch.from_numpy(host).to(self.device)
        lens = torch.tensor([s_.shape[0] for s_ in sigs], dtype=torch.int32, device=self.device)
        cap = 2 + ld // d.frame_shift
        mels = torch.empty((B, cap, d.mel), dtype=torch.float32, device=self.device)
        mel_len = torch.empty((B,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self.ctx.check(self.ctx.lib.gsttaco_mel_frontend(
                self.ctx.handle, _ptr(wav), _ptr(lens), B, ld, ctypes.c_float(float(top_db)), _ptr(mels), _ptr(mel_len), cap,
                self._stream()))
        n = int(mel_len.max().item())            # the one host sync: the output length is data dependent (trim)
        if int(mel_len.min().item()) < 1:
            raise ValueError("a reference wav is shorter than n_fft/2 samples after trimming (librosa.stft raises there)")
        return mels[:, :n + 1].contiguous(), mel_len

    def Inference(self, sentence_List, wav_List_for_GST=None, label=None, export=False, **kwargs):
        """reference Model.py:342-367.  ``wav_List_for_GST`` holds wav paths / 1-D sample arrays like the reference's,
        or precomputed mels [T, Mel_Dim].  The reference always starts its export thread; here ``export=True`` asks for
        it (it needs the CBHG vocoder and Griffin-Lim, which are off the mel-frame metric) and runs it synchronously."""
        print("Inference running...")
        pattern_Dict = self.feeder.Get_Inference_Pattern(sentence_List, wav_List_for_GST)
        if pattern_Dict is None:
            print("Inference fail.")
            return None
        if export:
            kwargs["with_vocoder"] = True
        out = self.Inference_Step(**pattern_Dict, **kwargs)
        self.synchronize()                     # the reference returns finished arrays; also where a give-up of this call surfaces
        if export:
            self.Export_Inference(sentence_List, out[0], out[1], out[2], out[3],
                                  label or datetime.now().strftime("%Y%m%d.%H%M%S"))
        return out

    def Inv_Spectrogram(self, spectrograms, frames=None, iters=None, power=1.5, ref_level_db=20.0, init_phase=None, seed=0):
        """Batched reference Audio.inv_spectrogram (Audio.py:23-27) on the GPU: spectrograms [B, T, Spectrogram_Dim] as
        Inference_Step returns them -> (wav [B, Frame_Shift*(T-1)] float32, wav_lengths [B]).  ``frames`` [B] limits the
        frames used per utterance; ``init_phase`` [B,T,Spectrogram_Dim] in [0,1) injects the random initial phases."""
        d = self.dims
        if not d.audio or not self.ctx.cfg.max_wav_samples:
            raise ValueError("the audio back end needs the Sound section of Hyper_Parameters and max_wav_seconds > 0")
        if not torch.cuda.is_available():
            raise capi.GstTacoError(-2, "no HIP device: the gfx950 kernels are the only compute path (no CPU fallback)")
        spec = self._dev(spectrograms, torch.float32)
        B, T = spec.shape[0], spec.shape[1]
        if iters is None:
            iters = int(self.hp_Dict.get("Vocoder_Taco1", {}).get("Griffin-Lim_Iter", 60))
        fr = self._dev(frames, torch.int32)
        ph = self._dev(init_phase, torch.float32)
        ld = max(1, d.frame_shift * (T - 1))
        wav = torch.empty((B, ld), dtype=torch.float32, device=self.device)
        lens = torch.empty((B,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self.ctx.check(self.ctx.lib.gsttaco_griffin_lim(
                self.ctx.handle, _ptr(spec), _ptr(fr), B, T, int(iters), ctypes.c_float(float(power)),
                ctypes.c_float(float(ref_level_db)), _ptr(ph), ctypes.c_uint64(int(seed)), _ptr(wav), _ptr(lens), ld,
                self._stream()))
        return wav, lens

    def Export_Inference(self, sentence_List, mel_List, stop_List, spectrogram_List, alignment_List, label, plot=True):
        """reference Model.py:369-427: per utterance a figure (Plot/<label>.IDX_<i>.PNG) and the Griffin-Lim wav of the
        spectrogram cut at the stop token (Wav/<label>.IDX_<i>.WAV) under Inference_Path.  Returns the wav paths."""
        from . import export
        root = self.hp_Dict["Inference_Path"]
        os.makedirs(os.path.join(root, "Plot"), exist_ok=True)
        os.makedirs(os.path.join(root, "Wav"), exist_ok=True)
        stops = np.asarray(stop_List.cpu() if torch.is_tensor(stop_List) else stop_List, dtype=np.float32)
        slice_idx = [export.stop_slice_index(s) for s in stops]
        frames = np.array([max(1, i) * self.dims.r for i in slice_idx], dtype=np.int32)           # Model.py:415
        wav, lens = self.Inv_Spectrogram(spectrogram_List, frames=frames)
        wav, lens = wav.cpu().numpy(), lens.cpu().numpy()
        to_np = lambda a: np.asarray(a.cpu() if torch.is_tensor(a) else a, dtype=np.float32)
        mels, specs, aligns = to_np(mel_List), to_np(spectrogram_List), to_np(alignment_List)
        paths = []
        for i, sentence in enumerate(sentence_List):
            if plot:
                try:
                    export.plot_inference(os.path.join(root, "Plot", "{}.IDX_{}.PNG".format(label, i)), sentence, mels[i],
                                          specs[i], aligns[i], stops[i], slice_idx[i])
                except ImportError:
                    plot = False                      # matplotlib is optional
            p = os.path.join(root, "Wav", "{}.IDX_{}.WAV".format(label, i))
            export.write_wav(p, wav[i, :lens[i]], self.dims.sample_rate)
            paths.append(p)
        return paths

    def Export_GST(self, wav_List, tag_List, gst_List, label):
        """reference Model.py:448-459"""
        from . import export
        gst = np.asarray(gst_List.cpu() if torch.is_tensor(gst_List) else gst_List)
        path = os.path.join(self.hp_Dict["Inference_Path"], "GST", "{}.GST.TXT".format(label))
        export.export_gst(path, wav_List, tag_List, gst)
        return path

    # ------------------------------------------------------------------ per-phase entry points (tests / profiling)
    def decode_plan(self, Tv):
        """(fused_front, fused_prenet0, lean) -- which variant of the decode step a Tv-token batch runs on
        (``gsttaco_decode_plan``).  The mixed-precision parity oracle needs ``fused_prenet0``."""
        self._require_ready()
        plan = (ctypes.c_int32 * 3)()
        self.ctx.check(self.ctx.lib.gsttaco_decode_plan(self.ctx.handle, int(Tv), plan))
        return bool(plan[0]), bool(plan[1]), bool(plan[2])

    def set_graph_policy(self, max_cached=8, capture_after=1):
        """hipGraph cache policy (``gsttaco_set_graph_policy``): at most ``max_cached`` graph executables are kept (LRU);
        a shape is captured at its ``capture_after``-th use and enqueued eagerly before.  Use ``capture_after=2`` when
        batch shapes vary from call to call (the reference's Feeder pads to the batch maximum, Feeder.py:175-180)."""
        self.ctx.check(self.ctx.lib.gsttaco_set_graph_policy(self.ctx.handle, int(max_cached), int(capture_after)))

    def synchronize(self):
        """Synchronises the current stream and raises GstTacoError if a hand-off wait of the persistent decode launch, of a persistent
        BiLSTM launch or of a fused decode-LSTM launch gave up since the last check (``gsttaco_synchronize``, the one place that clears the condition): the
        outputs of the calls since then are invalid and should be repeated -- the context runs the launch-per-step forms from
        then on.  ``Inference`` calls this before it returns."""
        with torch.cuda.device(self.device):
            self.ctx.check(self.ctx.lib.gsttaco_synchronize(self.ctx.handle, self._stream()))

    def last_message(self):
        """The library's last error or warning text for this context (``gsttaco_last_error``)."""
        return self.ctx.lib.gsttaco_last_error(self.ctx.handle).decode()

    def handoff_error(self):
        """Non-zero while a give-up is pending on this context, i.e. raised and not yet reported by ``synchronize`` (bit 0: fused
        decode-LSTM launch, bit 8: persistent BiLSTM, bit 16: persistent decode launch; ``gsttaco_debug_handoff_error``)."""
        out = ctypes.c_uint32(0)
        self.ctx.check(self.ctx.lib.gsttaco_debug_handoff_error(self.ctx.handle, ctypes.byref(out)))
        return int(out.value)

    def debug_counters(self):
        """(persistent BiLSTM launches this context has enqueued, 1 while the context still uses the persistent launch) --
        ``gsttaco_debug_counters``."""
        out = (ctypes.c_uint64 * 4)()
        self.ctx.check(self.ctx.lib.gsttaco_debug_counters(self.ctx.handle, out))
        return int(out[0]), int(out[1])

    def decode_counters(self):
        """(persistent decode launches this context has enqueued, 1 while the context still uses the persistent decode launch) --
        the whole decoder loop as ONE launch (``csrc/persist_decode.hip``; ``gsttaco_debug_counters`` out[2:4])."""
        out = (ctypes.c_uint64 * 4)()
        self.ctx.check(self.ctx.lib.gsttaco_debug_counters(self.ctx.handle, out))
        return int(out[2]), int(out[3])

    def graph_cache_size(self):
        return int(self.ctx.lib.gsttaco_graph_cache_size(self.ctx.handle))

    def debug_randomness(self, steps, B, Tv):
        """(prenet_masks [steps, 2, B, P], attn_noise [steps, B, Tv]) the last decode of that shape used, as NumPy arrays in
        the layout ``Inference_Step(prenet_masks=, attn_noise=)`` takes -- in throughput mode the tensors generated from the
        seed (``gsttaco_debug_randomness``; test support)."""
        import numpy as np
        self._require_ready()
        d = self.dims
        P0, P1 = d.prenet[0], d.prenet[1]
        if P0 != P1:
            raise ValueError("debug_randomness returns a stacked mask tensor: equal prenet sizes only")
        masks = np.empty((steps, B * (P0 + P1)), np.float32)
        noise = np.empty((steps, B, Tv), np.float32)
        self.ctx.check(self.ctx.lib.gsttaco_debug_randomness(self.ctx.handle, masks.ctypes.data_as(ctypes.c_void_p),
                                                             noise.ctypes.data_as(ctypes.c_void_p), int(steps), int(B), int(Tv)))
        return masks.reshape(steps, 2, B, P0), noise

    def encode(self, tokens, token_lengths=None):
        self._require_ready()
        tok = self._dev(tokens, torch.int32)
        tlen = self._dev(token_lengths, torch.int32)
        B, Tv = tok.shape
        enc = torch.empty((B, Tv, self.dims.enc_out), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self.ctx.check(self.ctx.lib.gsttaco_encode(self.ctx.handle, _ptr(tok), _ptr(tlen), B, Tv, _ptr(enc), self._stream()))
        return enc

    def decode(self, enc, gst=None, prenet_masks=None, attn_noise=None, seed=0, steps=None, token_lengths=None):
        self._require_ready()
        d = self.dims
        enc = self._dev(enc, torch.float32)
        gst = self._dev(gst, torch.float32)
        tlen = self._dev(token_lengths, torch.int32)
        B, Tv = enc.shape[0], enc.shape[1]
        S = d.steps if steps is None else int(steps)
        masks = self._dev(prenet_masks, torch.float32)
        noise = self._dev(attn_noise, torch.float32)
        pre = torch.empty((B, S * d.r, d.mel), dtype=torch.float32, device=self.device)
        stop = torch.empty((B, S), dtype=torch.float32, device=self.device)
        align = torch.empty((B, S, Tv), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self.ctx.check(self.ctx.lib.gsttaco_decode(
                self.ctx.handle, _ptr(enc), _ptr(gst), _ptr(tlen), _ptr(masks), _ptr(noise), ctypes.c_uint64(int(seed)),
                B, Tv, S, _ptr(pre), _ptr(stop), _ptr(align), self._stream()))
        return pre, stop, align

    def vocoder(self, mel):
        """Vocoder_Taco1 alone: mel [B,T,Mel_Dim] -> linear spectrogram [B,T,Spectrogram_Dim] (reference Taco2.py:234-260)."""
        self._require_ready()
        x = self._dev(mel, torch.float32)
        B, T = x.shape[0], x.shape[1]
        spec = torch.empty((B, T, self.dims.spec), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self.ctx.check(self.ctx.lib.gsttaco_vocoder(self.ctx.handle, _ptr(x), B, T, _ptr(spec), self._stream()))
        return spec

    def postnet(self, pre_mel):
        self._require_ready()
        pre = self._dev(pre_mel, torch.float32)
        B, T = pre.shape[0], pre.shape[1]
        mel = torch.empty_like(pre)
        with torch.cuda.device(self.device):
            self.ctx.check(self.ctx.lib.gsttaco_postnet(self.ctx.handle, _ptr(pre), B, T, _ptr(mel), self._stream()))
        return mel

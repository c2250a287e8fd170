"""Inference-side input conventions of the reference Feeder (reference Feeder.py:161-252).

Only what the hot path's caller needs: tokenise / pad, zero ``initial_mels``, and the
``mels_for_gst`` batch layout (a zero frame is prepended, lengths exclude it).  Reference audio is given
like in the reference as wav paths (or 1-D sample arrays): the wav -> mel front end
(Pattern_Generator.Mel_Generate) then runs on the GPU through the ``mel_frontend`` callable the
model installs, and ``mels_for_gst`` comes back as a device tensor.  Precomputed mel arrays ``[T, Mel_Dim]``
in [-Max_Abs_Mel, Max_Abs_Mel] are accepted too (no GPU involved).
"""
import numpy as np

from .hparams import load_token_dict


class Feeder:
    def __init__(self, hp, token_index_dict=None, mel_frontend=None):
        self.hp = hp
        self.token_Index_Dict = token_index_dict or load_token_dict(hp)
        self.mel_frontend = mel_frontend          # callable(wav_List, top_db) -> (mels_for_gst, mel_lengths_for_gst)

    @staticmethod
    def _is_mel(item):
        return not isinstance(item, (str, bytes)) and not hasattr(item, "__fspath__") and np.ndim(item) == 2

    def _from_wavs(self, wav_List, top_db):
        if self.mel_frontend is None:
            raise RuntimeError("wav inputs need the GPU mel front end (GST_Tacotron installs it); there is no CPU path")
        mels, lens = self.mel_frontend(wav_List, top_db)
        return {"mels_for_gst": mels, "mel_lengths_for_gst": lens}

    def Get_Inference_Pattern(self, sentence_List, mel_List_for_GST=None):
        """reference Feeder.py:161-227.  Out-of-vocabulary characters raise KeyError like the
        reference (Feeder.py:169); bad GST inputs print the reference's message and return None."""
        pattern_Count = len(sentence_List)
        sentence_List = [sentence.upper().strip() for sentence in sentence_List]          # :164
        token_List = [
            np.array([self.token_Index_Dict["<S>"]] + [self.token_Index_Dict[ch] for ch in s]
                     + [self.token_Index_Dict["<E>"]], dtype=np.int32)                     # :166-174
            for s in sentence_List]
        max_len = max(t.shape[0] for t in token_List)
        tokens = np.full((pattern_Count, max_len), self.token_Index_Dict["<E>"], dtype=np.int32)   # :177-180
        for i, t in enumerate(token_List):
            tokens[i, :t.shape[0]] = t
        mel_dim = self.hp["Sound"]["Mel_Dim"]
        pattern = {
            "tokens": tokens,
            "token_lengths": np.array([t.shape[0] for t in token_List], dtype=np.int32),
            "initial_mels": np.zeros((pattern_Count, 1, mel_dim), dtype=np.float32),        # :182-185
        }
        if self.hp["GST"]["Use"]:
            if mel_List_for_GST is None:
                print("GST is enabled, but no wav information.")                           # :197-199
                return None
            if len(mel_List_for_GST) not in (1, pattern_Count):
                print("The length of wav_List_for_GST must be 1 or same to the length of sentence_List "
                      "and wav_List_for_GST must be same.")                                 # :200-202
                return None
            items = list(mel_List_for_GST)
            if all(self._is_mel(m) for m in items):
                pattern.update(self.Get_Inference_GST_Pattern(items * (pattern_Count if len(items) == 1 else 1)))
            elif len(items) == 1:
                # one reference wav for every sentence: top_db 60, the mel is repeated (Feeder.py:204-207)
                one = self._from_wavs(items, 60)
                pattern["mels_for_gst"] = one["mels_for_gst"].expand(pattern_Count, -1, -1).contiguous()
                pattern["mel_lengths_for_gst"] = one["mel_lengths_for_gst"].expand(pattern_Count).contiguous()
            else:
                pattern.update(self._from_wavs(items, 15))        # several wavs: top_db 15 (Feeder.py:209)
        return pattern

    def Get_Inference_GST_Pattern(self, mel_List):
        """reference Feeder.py:229-252: zero-pad to the longest mel, PREPEND one zero frame.  Wav paths / sample
        arrays go through the GPU front end with top_db 60 (Feeder.py:232)."""
        mel_dim = self.hp["Sound"]["Mel_Dim"]
        if not all(self._is_mel(m) for m in mel_List):
            return self._from_wavs(list(mel_List), 60)
        mel_List = [np.asarray(m, dtype=np.float32) for m in mel_List]
        for m in mel_List:
            if m.ndim != 2 or m.shape[1] != mel_dim or m.shape[0] < 1:
                raise ValueError("reference mels must be [T>=1, {}] arrays".format(mel_dim))
        max_len = max(m.shape[0] for m in mel_List)
        mels = np.zeros((len(mel_List), max_len + 1, mel_dim), dtype=np.float32)
        for i, m in enumerate(mel_List):
            mels[i, 1:m.shape[0] + 1] = m
        return {"mels_for_gst": mels,
                "mel_lengths_for_gst": np.array([m.shape[0] for m in mel_List], dtype=np.int32)}
